import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
lib = _lib.init(0)
B, H, Cin, Cout = 256, 14, 1280, 640
x = torch.randn(B, H, H, Cin, device='cuda').bfloat16()
w = (torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.02)
g = torch.randn(B, H, H, Cout, device='cuda').bfloat16()
with torch.no_grad():
    for _ in range(3): y = ops.conv2d(x, w)
wd = w.detach().requires_grad_(True)
yw = ops.conv2d(x, wd)
for _ in range(3): torch.autograd.grad(yw, wd, g, retain_graph=True)
torch.cuda.synchronize()

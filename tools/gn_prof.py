"""GroupNorm forward / backward kernels on the U-Net's shapes, for `rocprofv3 --kernel-trace --stats` (kernel times without
host overhead): python tools/gn_prof.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
_lib.init(0)
B, reps = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 5
shapes = [(27, 320), (27, 640), (14, 640), (14, 1280), (7, 1280), (7, 2560), (4, 1280), (4, 2560)]
for H, C in shapes:
    x = torch.randn(B, H, H, C, device="cuda").bfloat16().requires_grad_(True)
    g = torch.ones(C, device="cuda", requires_grad=True); b = torch.zeros(C, device="cuda", requires_grad=True)
    for _ in range(reps):
        y, xp = ops.group_norm_split(x, g, b, 32, 1e-5, True)
        torch.autograd.grad((y, xp), (x, g, b), (torch.ones_like(y), torch.ones_like(xp)))
torch.cuda.synchronize()

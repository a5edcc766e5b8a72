"""Linear (1x1) layers of the attention blocks: TFLOP/s fwd / dgrad / wgrad at the U-Net's token counts."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
lib = _lib.init(0)
shapes = [(50176, 640, 640), (50176, 640, 1280), (50176, 640, 1920), (50176, 1280, 640), (12544, 1280, 1280), (12544, 1280, 2560),
          (12544, 1280, 3840), (12544, 2560, 1280), (4096, 1280, 1280), (4096, 1280, 3840), (8192, 1280, 2560)]
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
print("%-26s %8s %8s %8s" % ("M x K -> N", "fwd", "dgrad", "wgrad"))
for M, K, N in shapes:
    x = torch.randn(M, K, device='cuda').bfloat16().requires_grad_(True)
    w = (torch.randn(N, K, device='cuda') * 0.02).requires_grad_(True)
    b = torch.zeros(N, device='cuda', requires_grad=True)
    res = torch.randn(M, N, device='cuda').bfloat16()
    y = ops.linear(x, w, b, residual=res)
    g = torch.randn_like(y)
    fl = 2.0 * M * K * N
    with torch.no_grad():
        tf = timeit(lambda: ops.linear(x, w, b, residual=res))
    xd = x.detach().requires_grad_(True); yd = ops.linear(xd, w.detach(), None)
    td = timeit(lambda: torch.autograd.grad(yd, xd, g, retain_graph=True))
    wd = w.detach().requires_grad_(True); yw = ops.linear(x.detach(), wd, None)
    tw = timeit(lambda: torch.autograd.grad(yw, wd, g, retain_graph=True))
    print("%-26s %8.1f %8.1f %8.1f" % (f"{M} x {K} -> {N}", fl / tf / 1e12, fl / td / 1e12, fl / tw / 1e12))

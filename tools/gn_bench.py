import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
lib = _lib.init(0)
B = 256
shapes = [(27, 320), (27, 640), (14, 640), (14, 1280), (7, 1280), (7, 2560), (4, 1280), (4, 2560)]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
tf = tb = 0
for H, C in shapes:
    x = torch.randn(B, H, H, C, device='cuda').bfloat16().requires_grad_(True)
    g = torch.ones(C, device='cuda', requires_grad=True); b = torch.zeros(C, device='cuda', requires_grad=True)
    y = ops.group_norm(x, g, b, 32, 1e-5, True)
    dy = torch.randn_like(y)
    with torch.no_grad():
        t1 = timeit(lambda: ops.group_norm(x, g, b, 32, 1e-5, True))
    t2 = timeit(lambda: torch.autograd.grad(y, (x, g, b), dy, retain_graph=True))
    nbytes = B * H * H * C * 2
    print(f"{H}x{H}x{C}: fwd {t1*1e6:7.1f} us ({2*nbytes/t1/1e9:6.0f} GB/s)  bwd {t2*1e6:7.1f} us ({3*nbytes/t2/1e9:6.0f} GB/s)")
    tf += t1; tb += t2
print(f"sum fwd {tf*1e3:.2f} ms bwd {tb*1e3:.2f} ms")

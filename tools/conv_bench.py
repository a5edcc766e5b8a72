import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
lib = _lib.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shapes = [  # H, Cin, Cout, ks, stride
    (27, 320, 320, 3, 1), (27, 640, 320, 3, 1), (14, 640, 640, 3, 1), (14, 1280, 640, 3, 1), (7, 1280, 1280, 3, 1),
    (7, 2560, 1280, 3, 1), (4, 1280, 1280, 3, 1), (4, 2560, 1280, 3, 1), (27, 320, 640, 3, 2), (14, 640, 1280, 1, 1), (14, 640, 1920, 1, 1),
]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
print("%-28s %9s %9s %9s   (TFLOP/s fwd / dgrad / wgrad)" % ("shape", "fwd", "dgrad", "wgrad"))
for H, Cin, Cout, ks, st in shapes:
    x = torch.randn(B, H, H, Cin, device='cuda').bfloat16().requires_grad_(True)
    w = (torch.randn(Cout, Cin, ks, ks, device='cuda') * 0.02).requires_grad_(True)
    y = ops.conv2d(x, w, None, stride=st)
    Ho = y.shape[1]
    g = torch.randn_like(y)
    flops = 2.0 * B * Ho * Ho * Cout * Cin * ks * ks
    with torch.no_grad():
        tf = timeit(lambda: ops.conv2d(x, w, None, stride=st))
    # dgrad-only and wgrad-only via autograd with selective requires_grad
    xd = x.detach().requires_grad_(True); wn = w.detach()
    yd = ops.conv2d(xd, wn, None, stride=st)
    td = timeit(lambda: torch.autograd.grad(yd, xd, g, retain_graph=True))
    xn = x.detach(); wd = w.detach().requires_grad_(True)
    yw = ops.conv2d(xn, wd, None, stride=st)
    tw = timeit(lambda: torch.autograd.grad(yw, wd, g, retain_graph=True))
    print("%-28s %9.1f %9.1f %9.1f   ms %.3f %.3f %.3f" % (f"{H}x{H} {Cin}->{Cout} k{ks}s{st}", flops/tf/1e12, flops/td/1e12, flops/tw/1e12, tf*1e3, td*1e3, tw*1e3))

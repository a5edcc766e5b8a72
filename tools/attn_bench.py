import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
lib = _lib.init(0)
B = 256
cases = [(8, 196, 196, 80, True), (8, 196, 32, 80, False), (8, 49, 49, 160, True), (8, 49, 32, 160, False), (8, 16, 16, 160, True), (8, 16, 32, 160, False)]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
tf = tb = 0
for H, L, S, d, self_mode in cases:
    E = H * d
    if self_mode:
        q = torch.randn(B, L, 3 * E, device='cuda').bfloat16().requires_grad_(True); kv = None
        f = lambda: ops.attention_self(q, H, 0.05, 123)
    else:
        q = torch.randn(B, L, E, device='cuda').bfloat16().requires_grad_(True)
        kv = torch.randn(B, S, 2 * E, device='cuda').bfloat16().requires_grad_(True)
        f = lambda: ops.attention_cross(q, kv, H, 0.05, 123)
    o = f(); g = torch.randn_like(o)
    with torch.no_grad(): t1 = timeit(f)
    ins = (q,) if self_mode else (q, kv)
    t2 = timeit(lambda: torch.autograd.grad(o, ins, g, retain_graph=True))
    fl = 4.0 * B * H * L * S * d
    print(f"L={L:3d} S={S:3d} d={d:3d} {'self ' if self_mode else 'cross'}: fwd {t1*1e6:7.1f} us ({fl/t1/1e12:6.1f} TF)  bwd {t2*1e6:7.1f} us ({2.5*fl/t2/1e12:6.1f} TF)")
    tf += t1; tb += t2
print(f"sum fwd {tf*1e3:.2f} ms bwd {tb*1e3:.2f} ms")

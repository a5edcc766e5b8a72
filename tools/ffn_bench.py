"""The FFN's first GEMM (Linear + GELU, derivative saved) and its backward on the U-Net's three widths; HIP-event timing
of the autograd node: python tools/ffn_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
_lib.init(0)
B = 256
for HW, C in [(196, 640), (49, 1280), (16, 1280)]:
    M = B * HW
    x = torch.randn(M, C, device="cuda").bfloat16().requires_grad_(True)
    w1 = (torch.randn(2 * C, C, device="cuda") * 0.02).requires_grad_(True); b1 = torch.zeros(2 * C, device="cuda", requires_grad=True)
    w2 = (torch.randn(C, 2 * C, device="cuda") * 0.02).requires_grad_(True); b2 = torch.zeros(C, device="cuda", requires_grad=True)
    def fwd():
        return ops.ffn(x, w1, b1, w2, b2, 1.0, 0.05, 11, 12)
    y = fwd()
    if y is None:
        print("ops.ffn not found"); break
    g = torch.randn_like(y)
    def timed(fn, reps=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    with torch.no_grad():
        tf = timed(fwd)
    tb = timed(lambda: torch.autograd.grad(y, (x, w1, b1, w2, b2), g, retain_graph=True))
    fl = 2.0 * M * C * 2 * C * 2
    print(f"M {M} C {C}: fwd {tf:7.1f} us ({fl/tf/1e6:6.0f} TFLOP/s)  bwd {tb:7.1f} us ({2*fl/tb/1e6:6.0f} TFLOP/s)")

"""Pointwise layers of the U-Net at batch 256 on the persistent kernel (csrc/conv_pw.hip) against the per-tile kernel: HIP-event
time of back-to-back launches of each epilogue form through ops (forward only / data gradient via autograd off: direct launches).
python tools/pw_bench.py"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import _lib, ops
lib = _lib.init(0)
dev = "cuda"
dt = torch.bfloat16
reps = 10


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


shapes = [(50176, 640, 640), (50176, 640, 1280), (50176, 640, 1920), (50176, 1280, 640), (50176, 1920, 640), (12544, 1280, 1280),
          (12544, 1280, 3840), (12544, 1280, 2560), (12544, 2560, 1280), (8192, 1280, 2560)]
if len(sys.argv) > 1 and sys.argv[1] == "longk":      # K as long as a 3x3 layer's: is the persistent form's K loop itself as fast as the per-tile kernel's?
    shapes = [(50176, 5760, 640), (12544, 11520, 1280), (50176, 2560, 640)]
print("%-22s %-14s %9s %9s %7s" % ("M,K,N", "form", "tiled us", "pw us", "gain"))
tot = [0.0, 0.0]
for M, K, N in shapes:
    x = torch.randn(M, K, device=dev).to(dt)
    res = torch.randn(M, N, device=dev).to(dt)
    w = torch.randn(N, K, device=dev) * math.sqrt(1.0 / K)
    b = torch.randn(N, device=dev)
    wf, _ = ops.WeightCache.get(w, dt, False)
    y = torch.empty(M, N, device=dev, dtype=dt)
    pre = torch.empty(M, N, device=dev, dtype=dt)
    geo = (M, 1, 1, 1, 1, 1, 1, 0)
    forms = {
        "bias": lambda: ops._conv_launch(lib, dt, x, K, wf, 0, y, N, geo, K, N, bias=b),
        "bias+res": lambda: ops._conv_launch(lib, dt, x, K, wf, 0, y, N, geo, K, N, bias=b, residual=res, ld_res=N, alpha=0.7),
        "drop+res": lambda: ops._conv_launch(lib, dt, x, K, wf, 0, y, N, geo, K, N, bias=b, residual=res, ld_res=N, alpha=0.6, drop_p=0.05, seed=5),
        "gelu+drop+dact": lambda: ops._conv_launch(lib, dt, x, K, wf, 0, y, N, geo, K, N, bias=b, preact=pre, act=ops.ACT_GELU, drop_p=0.05, seed=5, flags=_lib.CONV_SAVE_DACT),
        "dmul": lambda: ops._conv_launch(lib, dt, x, K, wf, 0, y, N, geo, K, N, transposed=True, dact_u=res, ld_dact=N, flags=_lib.CONV_DACT_MUL),
        "plain": lambda: ops._conv_launch(lib, dt, x, K, wf, 0, y, N, geo, K, N, transposed=True),
    }
    for name, fn in forms.items():
        _lib.check(lib.psg_conv_set_pw(0), "pw")
        t0 = timed(fn)
        _lib.check(lib.psg_conv_set_pw(1), "pw")
        c = lib.psg_conv_pw_launches()
        t1 = timed(fn)
        took = lib.psg_conv_pw_launches() > c
        tot[0] += t0; tot[1] += t1
        fl = 2.0 * M * K * N
        print("%-22s %-14s %9.1f %9.1f %6.1f%%  %6.0f -> %6.0f TFLOP/s%s" % (f"{M},{K},{N}", name, t0, t1, 100 * (t0 / t1 - 1), fl / t0 / 1e6, fl / t1 / 1e6, "" if took else "  (not taken)"))
print("sum: tiled %.0f us, pw %.0f us" % tuple(tot))

"""Bilinear upsample forward / backward on the U-Net's three shapes at batch 256 (bf16): us and GB/s (read + write once)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import _lib
lib = _lib.init(0)
B, reps = 256, 20
st = torch.cuda.current_stream().cuda_stream


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = [0.0, 0.0]
for hi, ho, C in ((4, 7, 1280), (7, 14, 1280), (14, 27, 640)):
    x = torch.randn(B, hi, hi, C, device="cuda").bfloat16(); y = torch.empty(B, ho, ho, C, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn_like(y); dx = torch.empty_like(x)
    p = lambda t: t.data_ptr()
    f = lambda: _lib.check(lib.psg_upsample_bilinear_fwd(p(x), C, p(y), C, B, hi, hi, ho, ho, C, _lib.PSG_BF16, st))
    b = lambda: _lib.check(lib.psg_upsample_bilinear_bwd(p(dy), C, p(dx), C, B, hi, hi, ho, ho, C, _lib.PSG_BF16, st))
    nb = (x.numel() + y.numel()) * 2
    tf, tb = timed(f), timed(b)
    tot[0] += tf; tot[1] += tb
    print(f"{hi}->{ho} x{C}: fwd {tf:7.1f} us ({nb/tf/1e3:6.0f} GB/s)  bwd {tb:7.1f} us ({nb/tb/1e3:6.0f} GB/s)")
print("sum us: fwd %.0f bwd %.0f" % tuple(tot))

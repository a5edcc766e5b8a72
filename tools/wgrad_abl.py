"""Ablation timing of one wgrad layer: python tools/wgrad_abl.py H Cin Cout ks  (library via PSG_LIB_PATH)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
lib = _lib.init(0)
B = 256
for H, Cin, Cout, ks in [(14, 640, 640, 3), (7, 1280, 1280, 3), (14, 1280, 640, 3)]:
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    dy = torch.randn(B, H, H, Cout, device="cuda").bfloat16()
    dw = torch.empty((Cout, Cin, ks, ks), device="cuda").contiguous(memory_format=torch.channels_last)
    geom = (B, H, H, H, H, ks, 1, 1 if ks == 3 else 0)
    f = lambda: ops._wgrad_launch(lib, torch.bfloat16, x, Cin, dy, Cout, dw, geom, Cin, Cout)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print("%s %dx%d %d->%d k%d: %.1f us %.0f TFLOP/s" % (os.environ.get("PSG_LIB_PATH", "full")[-12:], H, H, Cin, Cout, ks, us, 2.0 * B * H * H * Cout * Cin * ks * ks / us / 1e6))

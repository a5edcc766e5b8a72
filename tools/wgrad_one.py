"""One wgrad layer, timed: python tools/wgrad_one.py H Cin Cout ks [B]  (plan knobs via PSG_WGRAD_* env)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
lib = _lib.init(0)
H, Cin, Cout, ks = (int(v) for v in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 256
x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
dy = torch.randn(B, H, H, Cout, device="cuda").bfloat16()
dw = torch.empty((Cout, Cin, ks, ks), device="cuda").contiguous(memory_format=torch.channels_last)
db = torch.empty(Cout, device="cuda")
geom = (B, H, H, H, H, ks, 1, 1 if ks == 3 else 0)
f = lambda: ops._wgrad_launch(lib, torch.bfloat16, x, Cin, dy, Cout, dw, geom, Cin, Cout, dbias=db)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
print("%dx%d %d->%d k%d splits=%s wide=%s: %.1f us %.0f TFLOP/s" % (H, H, Cin, Cout, ks, os.environ.get("PSG_WGRAD_SPLITS", "auto"), os.environ.get("PSG_WGRAD_WIDE", "1"), us, 2.0 * B * H * H * Cout * Cin * ks * ks / us / 1e6))

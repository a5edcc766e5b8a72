"""Forward 1x1 (pointwise) layers alone, with a bias, no_grad: us per launch.  Ablation builds via PSG_LIB_PATH."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
lib = _lib.init(0)
B = 256
tag = os.environ.get("PSG_LIB_PATH", "full").split("/")[-2] if os.environ.get("PSG_LIB_PATH") else "full"
out = []
for H, Cin, Cout in [(14, 640, 640), (14, 640, 1920), (7, 1280, 1280), (7, 1280, 3840), (4, 1280, 1280), (14, 1920, 640)]:
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.02)
    b = torch.randn(Cout, device="cuda")
    with torch.no_grad():
        f = lambda: ops.conv2d(x, w, b)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    out.append("%dx%d %d->%d %.1f us (%.0f TF)" % (H, H, Cin, Cout, us, 2.0 * B * H * H * Cin * Cout / us / 1e6))
print(tag, " | ".join(out))

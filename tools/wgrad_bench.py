"""Weight-gradient kernel alone on the U-Net's layer shapes (batch 256, bf16, OHWI gradient), through ops._wgrad_launch:
us per launch and TFLOP/s.  A/B a kernel change inside ONE gpurun call: run it twice with PSG_WGRAD_WIDE=0 / 1."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
lib = _lib.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
shapes = [  # H, Cin, Cout, ks, launches per step
    (14, 640, 640, 3, 6), (7, 1280, 1280, 3, 7), (27, 320, 320, 3, 6), (7, 2560, 1280, 3, 2), (14, 1280, 640, 3, 3), (27, 640, 320, 3, 3),
    (4, 1280, 1280, 3, 8), (4, 2560, 1280, 3, 2), (14, 640, 640, 1, 12), (7, 1280, 1280, 1, 12), (14, 640, 1920, 1, 4), (7, 1280, 3840, 1, 4),
    (14, 640, 1280, 1, 4), (14, 1280, 640, 1, 4), (7, 1280, 2560, 1, 4), (7, 2560, 1280, 1, 4), (4, 1280, 1280, 1, 15), (4, 1280, 3840, 1, 5),
]
tot = 0.0
print("%-26s %9s %9s %9s" % ("shape", "us", "TFLOP/s", "ms/step"))
for H, Cin, Cout, ks, n in shapes:
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    dy = torch.randn(B, H, H, Cout, device="cuda").bfloat16()
    dw = torch.empty((Cout, Cin, ks, ks), device="cuda").contiguous(memory_format=torch.channels_last)
    db = torch.empty(Cout, device="cuda")
    geom = (B, H, H, H, H, ks, 1, 1 if ks == 3 else 0)
    f = lambda: ops._wgrad_launch(lib, torch.bfloat16, x, Cin, dy, Cout, dw, geom, Cin, Cout, dbias=db)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * B * H * H * Cout * Cin * ks * ks
    tot += us * n / 1e3
    print("%-26s %9.1f %9.1f %9.3f" % (f"{H}x{H} {Cin}->{Cout} k{ks}", us, fl / us / 1e6, us * n / 1e3))
print("sum ms/step %.2f" % tot)

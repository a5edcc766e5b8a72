import os, sys, subprocess
shapes = "(14, 640, 640, 1), (7, 1280, 1280, 1), (14, 640, 1920, 1), (7, 1280, 3840, 1), (14, 640, 1280, 1), (14, 1280, 640, 1), (7, 1280, 2560, 1), (7, 2560, 1280, 1), (4, 1280, 1280, 1), (4, 1280, 3840, 1)"
code = '''
import os, sys, torch
sys.path.insert(0, "%s")
from pokemon_sprite_generator_amd import ops, _lib
lib = _lib.init(0)
B = 256
out = []
for H, Cin, Cout, ks in [%s]:
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    dy = torch.randn(B, H, H, Cout, device="cuda").bfloat16()
    dw = torch.empty((Cout, Cin, ks, ks), device="cuda").contiguous(memory_format=torch.channels_last)
    db = torch.empty(Cout, device="cuda")
    geom = (B, H, H, H, H, ks, 1, 0)
    f = lambda: ops._wgrad_launch(lib, torch.bfloat16, x, Cin, dy, Cout, dw, geom, Cin, Cout, dbias=db)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    out.append("%%6.1f" %% (e0.elapsed_time(e1) / 20 * 1e3))
print(" ".join(out))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), shapes)
print("shapes:", shapes)
for s in ["", "4", "6", "8", "10", "12", "16", "20", "24", "32", ""]:
    env = dict(os.environ)
    if s: env["PSG_WGRAD_SPLITS"] = s
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("splits %4s:" % (s or "auto"), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])

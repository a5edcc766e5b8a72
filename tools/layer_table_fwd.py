"""Per-layer table of the EVAL forward (the sampler's denoising step): every distinct conv_fwd launch shape, count per step,
us per launch (re-launched in place) and TFLOP/s, sorted by total time.  python tools/layer_table_fwd.py [B]"""
import sys, collections, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokemon_sprite_generator_amd as psg
from pokemon_sprite_generator_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = _lib.init(0)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
unet = psg.UNet(compute_dtype=torch.bfloat16).to(dev).eval()
lat, txt = torch.randn(B, 8, 27, 27, device=dev), torch.randn(B, 32, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
with torch.no_grad():
    for _ in range(2):
        unet(lat, t, txt)
torch.cuda.synchronize()
rec = collections.OrderedDict()
orig_conv = ops._conv_launch
REPS = 10


def conv_hook(lib_, dtype, x, ldx, w, ldw, y, ldy, geom, Cin, Cout, transposed=False, **kw):
    Bq, Hi, Wi, Ho, Wo, ks, stride, pad = geom
    flops = 2.0 * Bq * Ho * Wo * Cout * Cin * ks * ks
    key = (geom, Cin, Cout, ldx, ldy)
    fn = lambda: orig_conv(lib_, dtype, x, ldx, w, ldw, y, ldy, geom, Cin, Cout, transposed=transposed, **kw)
    fn()
    if key in rec:
        rec[key][1] += 1
        return
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    rec[key] = [e0.elapsed_time(e1) / REPS, 1, flops]


ops._conv_launch = conv_hook
with torch.no_grad():
    unet(lat, t, txt)
torch.cuda.synchronize()
rows = sorted(rec.items(), key=lambda kv: -kv[1][0] * kv[1][1])
tot = 0.0
print("%-44s %5s %9s %9s %8s" % ("B,Hi,Wi,Ho,Wo,k,s,p Cin->Cout", "n", "us/launch", "ms/step", "TFLOP/s"))
for (geom, Cin, Cout, ldx, ldy), (ms, n, fl) in rows:
    tot += ms * n
    if ms * n > 0.04:
        print("%-44s %5d %9.1f %9.3f %8.1f" % (f"M{geom[0]*geom[3]*geom[4]} " + ",".join(map(str, geom[1:])) + f" {Cin}->{Cout}", n, ms * 1e3, ms * n, fl / ms / 1e9))
print("total conv ms/step %.2f in %d launches" % (tot, sum(v[1] for v in rec.values())))

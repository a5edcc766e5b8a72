"""Per-layer GEMM table of the real train step: every distinct conv_fwd / dgrad / wgrad launch shape, its count per
step, time per launch (re-launched in place, 5 reps) and TFLOP/s; sorted by total time."""
import sys, collections, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokemon_sprite_generator_amd as psg
from pokemon_sprite_generator_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.init(0)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
unet = psg.UNet(compute_dtype=torch.bfloat16).to(dev)
st = psg.DiffusionStepper(unet, psg.NoiseScheduler(), distributed=False)
lat, txt = torch.randn(B, 8, 27, 27, device=dev), torch.randn(B, 32, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
for _ in range(2):
    st.train_step(lat, txt, t)
torch.cuda.synchronize()
rec = collections.OrderedDict()
orig_conv, orig_wgrad = ops._conv_launch, ops._wgrad_launch
REPS = 5


def timed(key, flops, fn):
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


def conv_hook(lib_, dtype, x, ldx, w, ldw, y, ldy, geom, Cin, Cout, transposed=False, **kw):
    Bq, Hi, Wi, Ho, Wo, ks, stride, pad = geom
    flops = 2.0 * Bq * Ho * Wo * Cout * Cin * ks * ks if not transposed else 2.0 * Bq * Hi * Wi * Cout * Cin * ks * ks
    if transposed and stride == 2:
        flops = 2.0 * Bq * Hi * Wi * Cout * Cin * ks * ks     # useful flops: dY pixels x taps
    key = ("dgrad" if transposed else "fwd", geom, Cin, Cout, ldx, ldy)
    timed(key, flops, lambda: orig_conv(lib_, dtype, x, ldx, w, ldw, y, ldy, geom, Cin, Cout, transposed=transposed, **kw))


def wgrad_hook(lib_, dtype, x, ldx, dy, lddy, dw, geom, Cin, Cout, accumulate=False, **kw):
    Bq, Hi, Wi, Ho, Wo, ks, stride, pad = geom
    flops = 2.0 * Bq * Ho * Wo * Cout * Cin * ks * ks
    key = ("wgrad", geom, Cin, Cout, ldx, lddy)
    timed(key, flops, lambda: orig_wgrad(lib_, dtype, x, ldx, dy, lddy, dw, geom, Cin, Cout, accumulate=accumulate, **kw))


ops._conv_launch, ops._wgrad_launch = conv_hook, wgrad_hook
st.train_step(lat, txt, t)
torch.cuda.synchronize()
rows = sorted(rec.items(), key=lambda kv: -kv[1][0] * kv[1][1])
tot = collections.defaultdict(float)
print("%-6s %-34s %5s %9s %9s %8s" % ("kind", "B,Hi,Wi,Ho,Wo,k,s,p Cin->Cout", "n", "us/launch", "ms/step", "TFLOP/s"))
for (kind, geom, Cin, Cout, ldx, ldy), (ms, n, fl) in rows:
    tot[kind] += ms * n
    if ms * n > 0.25:
        print("%-6s %-40s %5d %9.1f %9.3f %8.1f" % (kind, f"M{geom[0]*geom[3]*geom[4]} " + ",".join(map(str, geom[1:])) + f" {Cin}->{Cout}" + (f" ldx{ldx}" if ldx != Cin else ""), n, ms * 1e3, ms * n, fl / ms / 1e9))
print({k: round(v, 2) for k, v in tot.items()})

"""Which Python lines launch torch's own copy / add / cat kernels in a train step (torch.profiler, CPU+GPU, with_stack)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokemon_sprite_generator_amd as psg
from pokemon_sprite_generator_amd import _lib
_lib.init(0)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
unet = psg.UNet(compute_dtype=torch.bfloat16).to(dev)
st = psg.DiffusionStepper(unet, psg.NoiseScheduler(), distributed=False)
lat, txt = torch.randn(B, 8, 27, 27, device=dev), torch.randn(B, 32, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
for _ in range(2): st.train_step(lat, txt, t)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    st.train_step(lat, txt, t)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    ks = [k.name for k in (e.kernels or [])]
    if not ks or not e.name.startswith("aten::"):
        continue
    if any(("at::native" in k or "copyBuffer" in k) for k in ks):
        stack = [f for f in (e.stack or []) if "pokemon_sprite_generator_amd" in f or "autograd" in f][:3]
        cnt[(e.name, ks[0][:50], " <- ".join(s.split("/")[-1][:60] for s in stack))] += 1
for (n, k, stk), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c:4d} {n:18s} {k:52s} {stk}")

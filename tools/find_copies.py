"""Which Python lines issue the small device-to-device copies of a train step (torch.profiler, with_stack)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokemon_sprite_generator_amd as psg
from pokemon_sprite_generator_amd import _lib
_lib.init(0)
dev = torch.device("cuda", 0)
B = 64
unet = psg.UNet(compute_dtype=torch.bfloat16).to(dev)
st = psg.DiffusionStepper(unet, psg.NoiseScheduler(), distributed=False)
lat, txt = torch.randn(B, 8, 27, 27, device=dev), torch.randn(B, 32, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
for _ in range(2): st.train_step(lat, txt, t)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    st.train_step(lat, txt, t)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::cat"):
        shp = ""
        stack = [f for f in (e.stack or []) if "pokemon_sprite_generator_amd" in f][:2]
        cnt[(e.name, shp, " <- ".join(s.split("/")[-1] for s in stack))] += 1
for (n, shp, stk), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{c:4d} {n:16s} {shp:62s} {stk}")

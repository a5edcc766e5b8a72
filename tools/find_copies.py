"""Which Python lines launch torch's own copy / add / fill / cat work in a train step (torch.profiler, CPU+GPU, with_stack):
every aten:: op that runs a kernel or a memcpy, grouped by (op, shapes, innermost package frames)."""
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
LEAF = ("aten::copy_", "aten::add_", "aten::add", "aten::fill_", "aten::zero_", "aten::cat", "aten::mul_", "aten::mul", "aten::sqrt",
        "aten::bitwise_and", "aten::bitwise_or", "aten::bitwise_or_", "aten::sum", "aten::normal_", "aten::random_", "aten::clamp")
for e in prof.events():
    if e.name not in LEAF:
        continue
    stack = [f for f in (e.stack or []) if "pokemon_sprite_generator_amd" in f or "tools/" in f][:3]
    shapes = ""
    cnt[(e.name, shapes, " <- ".join(s.split("/")[-1][:48] for s in stack))] += 1
for (n, sh, stk), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:50]:
    print(f"{c:4d} {n:16s} {sh:42s} {stk}")

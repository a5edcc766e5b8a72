"""Host time to ENQUEUE one train step (no synchronisation inside the loop) against the GPU time of the step: how far the
step is from being launch-bound.  python tools/host_time.py [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokemon_sprite_generator_amd as psg
from pokemon_sprite_generator_amd import _lib
_lib.init(0)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
unet = psg.UNet(compute_dtype=torch.bfloat16).to(dev)
st = psg.DiffusionStepper(unet, psg.NoiseScheduler(), distributed=False)
lat, txt = torch.randn(B, 8, 27, 27, device=dev), torch.randn(B, 32, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
for _ in range(3): st.train_step(lat, txt, t)
torch.cuda.synchronize()
n = 8
t0 = time.perf_counter()
for _ in range(n): st.train_step(lat, txt, t)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"batch {B}: host enqueue {1e3*(t1-t0)/n:.1f} ms/step, wall {1e3*(t2-t0)/n:.1f} ms/step")

"""Whole train step as a hipGraph replay against the eager step: python tools/graph_step_bench.py [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokemon_sprite_generator_amd as psg
from pokemon_sprite_generator_amd import _lib, ops
_lib.init(0)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
st = psg.DiffusionStepper(psg.UNet(compute_dtype=torch.bfloat16).to(dev), psg.NoiseScheduler(), distributed=False)
lat, txt = torch.randn(B, 8, 27, 27, device=dev), torch.randn(B, 32, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
n = 10
for _ in range(3): st.train_step(lat, txt, t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): st.train_step(lat, txt, t)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"batch {B}: eager        {1e3*(t1-t0)/n:.2f} ms/step")
gs = st.capture_train_step(lat, txt, t, warmup=2)
for _ in range(2): gs.run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): gs.run()
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"batch {B}: graph replay {1e3*(t1-t0)/n:.2f} ms/step")

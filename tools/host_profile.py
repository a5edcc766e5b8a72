"""cProfile of the host side of train steps at a small batch (where the step is launch-bound): python tools/host_profile.py"""
import os, sys, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokemon_sprite_generator_amd as psg
from pokemon_sprite_generator_amd import _lib
_lib.init(0)
dev = torch.device("cuda", 0)
B = 8
unet = psg.UNet(compute_dtype=torch.bfloat16).to(dev)
st = psg.DiffusionStepper(unet, psg.NoiseScheduler(), distributed=False)
lat, txt = torch.randn(B, 8, 27, 27, device=dev), torch.randn(B, 32, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
for _ in range(3): st.train_step(lat, txt, t)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): st.train_step(lat, txt, t)
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr).sort_stats("tottime")
ps.print_stats(28)

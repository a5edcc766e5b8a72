import os, sys, torch
sys.path.insert(0, '/root/repo')
import pokemon_sprite_generator_amd as psg
dev = torch.device('cuda', 0)
torch.manual_seed(0)
unet = psg.UNet(compute_dtype=torch.bfloat16).to(dev)
st = psg.DiffusionStepper(unet, psg.NoiseScheduler(), distributed=False)
B = 256
lat, txt = torch.randn(B, 8, 27, 27, device=dev), torch.randn(B, 32, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
st.train_step(lat, txt, t)
torch.cuda.synchronize()

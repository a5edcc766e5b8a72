"""End-to-end training sanity on one small fixed dataset: the loss on a FIXED (t, noise) probe must fall while the stepper
trains on fresh (t, noise) draws.  python tools/train_sanity.py [steps] [lr]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokemon_sprite_generator_amd as psg
from pokemon_sprite_generator_amd import _lib
_lib.init(0)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = 8
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 5e-4
unet = psg.UNet(compute_dtype=torch.bfloat16).to(dev)
st = psg.DiffusionStepper(unet, psg.NoiseScheduler(), distributed=False)
lat, txt = torch.randn(B, 8, 27, 27, device=dev).clamp(-3, 3), torch.randn(B, 32, 256, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
tp = torch.randint(0, 1000, (B,), device=dev, generator=g)
np_ = torch.randn(lat.shape, device=dev, generator=g)
p0 = st.params.flat.clone()
for i in range(steps + 1):
    if i % 250 == 0:
        l, _ = st.eval_loss(lat, txt, tp, noise=np_)
        torch.cuda.synchronize()
        d = (st.params.flat - p0).norm().item()
        print(f"step {i:4d} probe loss {float(l.item()):.4f}  |dparam| {d:.3f}  opt steps {st.steps_done()}")
    if os.environ.get("FIXED", "0") != "0":          # overfit the probe batch itself
        t, noise = tp, np_
    else:
        t = torch.randint(0, 1000, (B,), device=dev, generator=g)
        noise = torch.randn(lat.shape, device=dev, generator=g)
    out = st.train_step(lat, txt, t, noise=noise, lr=lr)
    if i % 250 == 0:
        print(f"          train loss {float(out['loss'].item()):.4f} grad_norm {float(out['grad_norm'].item()):.4f} flag {int(out['nan_flag'].item())}")

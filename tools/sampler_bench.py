"""ddpm_sample latency: hipGraph replay of the step vs eager launches (bf16, 20 fast steps)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pokemon_sprite_generator_amd as psg
dev = torch.device('cuda', 0)
torch.manual_seed(0)
u = psg.UNet(compute_dtype=torch.bfloat16).to(dev)
st = psg.DiffusionStepper(u, psg.NoiseScheduler(), distributed=False)
for B in (1, 4, 16):
    text = torch.randn(B, 32, 256, device=dev)
    for mode in (True, False):
        st.sample(text, B, True, use_graph=mode)            # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st.sample(text, B, True, use_graph=mode)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"B={B:3d} {'graph' if mode else 'eager'}: {dt*1e3:8.1f} ms for 20 steps ({dt/20*1e3:6.2f} ms/step)")

"""GroupNorm kernels through the C ABI, timed with HIP events over back-to-back launches (no autograd / host overhead in the
figure): python tools/gn_direct.py [reps].  Columns: forward (+SiLU), backward (+SiLU), backward with the fused bypass
gradient; GB/s counts 2 / 3 / 4 tensor passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import _lib
lib = _lib.init(0)
B, reps = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [(27, 320), (27, 640), (14, 640), (14, 1280), (7, 1280), (7, 2560), (4, 1280), (4, 2560)]
st = torch.cuda.current_stream().cuda_stream


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = [0.0, 0.0, 0.0]
for H, C in shapes:
    HW, G = H * H, 32
    x = torch.randn(B, HW, C, device="cuda").bfloat16()
    dy = torch.randn_like(x); res = torch.randn_like(x)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    gm = torch.ones(C, device="cuda"); bt = torch.zeros(C, device="cuda")
    mean = torch.empty(B * G, device="cuda"); rstd = torch.empty(B * G, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    wsf = torch.empty(max(lib.psg_groupnorm_fwd_workspace_bytes(B, C), 16), dtype=torch.uint8, device="cuda")
    wsb = torch.empty(max(lib.psg_groupnorm_bwd_workspace_bytes(B, C), 16), dtype=torch.uint8, device="cuda")
    p = lambda t: t.data_ptr()
    f = lambda: _lib.check(lib.psg_groupnorm_fwd(p(x), C, p(y), C, p(gm), p(bt), p(mean), p(rstd), B, HW, C, G, 1e-5, 1, _lib.PSG_BF16, p(wsf), st))
    b0 = lambda: _lib.check(lib.psg_groupnorm_bwd_res(p(dy), C, p(x), C, p(gm), p(bt), p(mean), p(rstd), None, 0, p(dx), C, p(dg), p(db), B, HW, C, G, 1, 0, _lib.PSG_BF16, p(wsb), st))
    b1 = lambda: _lib.check(lib.psg_groupnorm_bwd_res(p(dy), C, p(x), C, p(gm), p(bt), p(mean), p(rstd), p(res), C, p(dx), C, p(dg), p(db), B, HW, C, G, 1, 0, _lib.PSG_BF16, p(wsb), st))
    n = B * HW * C * 2
    t = [timed(f), timed(b0), timed(b1)]
    for i in range(3): tot[i] += t[i]
    print(f"{H}x{H}x{C}: fwd {t[0]:7.1f} us ({2*n/t[0]/1e3:6.0f} GB/s)  bwd {t[1]:7.1f} us ({3*n/t[1]/1e3:6.0f} GB/s)  bwd+res {t[2]:7.1f} us ({4*n/t[2]/1e3:6.0f} GB/s)")
print("sum us: fwd %.0f bwd %.0f bwd+res %.0f" % tuple(tot))

"""One Linear / 1x1 GEMM through the C ABI with its epilogue variants, HIP-event timing of back-to-back launches:
    python tools/gemm_direct.py M K N [reps]      (PSG_CONV_TILE=0..4 pins the tile, PSG_EPI_KINDS=0 the run-time epilogue)"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokemon_sprite_generator_amd import ops, _lib
from pokemon_sprite_generator_amd._lib import ConvDesc
lib = _lib.init(0)
M, K, N = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.02)
wf, wd = ops.WeightCache.get(w, torch.bfloat16, True)
bias = torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda").bfloat16()
sav = torch.randn(M, N, device="cuda").bfloat16()
y = torch.empty(M, N, device="cuda").bfloat16()
pre = torch.empty(M, N, device="cuda").bfloat16()
dx = torch.empty(M, K, device="cuda").bfloat16()
g = torch.randn(M, N, device="cuda").bfloat16()


def desc(transposed=False, **kw):
    d = ConvDesc()
    d.dtype = _lib.PSG_BF16
    cin, cout = (N, K) if transposed else (K, N)
    d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = M, 1, 1, cin, 1, 1, cout
    d.ksize, d.stride, d.pad, d.transposed, d.act = 1, 1, 0, int(transposed), kw.get("act", 0)
    d.alpha, d.drop_p, d.drop_seed, d.flags = 1.0, kw.get("drop", 0.0), 1234, kw.get("flags", 0)
    if transposed:
        d.ldx, d.ldy, d.ldw = N, K, wd.shape[1] if wd.dim() > 1 else 0
        d.x, d.w, d.y = g.data_ptr(), wd.data_ptr(), dx.data_ptr()
    else:
        d.ldx, d.ldy, d.ldw = K, N, 0
        d.x, d.w, d.y = x.data_ptr(), wf.data_ptr(), y.data_ptr()
    if kw.get("bias"): d.bias = bias.data_ptr()
    if kw.get("res"): d.residual, d.ld_residual = res.data_ptr(), N
    if kw.get("pre"): d.preact, d.ld_preact = pre.data_ptr(), N
    if kw.get("dact"): d.dact_u, d.ld_dact = sav.data_ptr(), (K if transposed else N)
    return d


def timed(d):
    fn = lambda: _lib.check(lib.psg_conv_fwd(C.byref(d), st))
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fl = 2.0 * M * K * N
for name, d in [("plain", desc()), ("bias", desc(bias=1)), ("bias+res", desc(bias=1, res=1)), ("bias+res+drop", desc(bias=1, res=1, drop=0.05)),
                ("bias+gelu", desc(bias=1, act=_lib.ACT_GELU if hasattr(_lib, "ACT_GELU") else 2)),
                ("bias+gelu+drop+save", desc(bias=1, act=_lib.ACT_GELU if hasattr(_lib, "ACT_GELU") else 2, drop=0.05, pre=1, flags=1)),
                ("dgrad plain", desc(True))]:
    t = timed(d)
    print(f"{M}x{K}->{N} {name:22s} {t:7.1f} us  {fl/t/1e6:6.0f} TFLOP/s")

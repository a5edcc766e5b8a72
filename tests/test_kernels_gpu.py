"""-m gpu: every HIP kernel, through the C ABI, against the CPU oracle / torch fp32 reference
on the same seeded inputs.  Bit-exact for index/table work, tolerance stated per test for fp."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as O
from tests.util import TOL, from_cl, h, maxrel, rel_l2, to_cl

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def _q(x, dtype):
    """Round an fp32 CPU tensor to the compute dtype (so the reference sees the same inputs)."""
    return x.to(dtype).float()


@pytest.fixture(scope="module")
def psg():
    import pokemon_sprite_generator_amd as m
    from pokemon_sprite_generator_amd import _lib
    _lib.init(0)
    return m


def _fixture_scheduler(psg, golden):
    """NoiseScheduler carrying the tables of the committed fixture.  torch's vectorised CPU cos/sqrt differ in
    the last bit between host CPUs, so the REFERENCE's own tables are host-dependent; fixtures made in the
    authoring container are compared with the fixture's tables loaded, and host-built tables are compared
    with the oracle run on the same host."""
    g = golden("schedule.npz")
    s = psg.NoiseScheduler()
    for n in ("betas", "alphas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
        setattr(s, n, torch.from_numpy(g["cos_" + n]).clone())
    return s


# ---------------------------------------------------------------- a-2 add_noise (bit-exact)
def test_noise_add_bit_exact(psg, golden):
    g = golden("add_noise.npz")
    x0 = torch.clamp(h((6, 8, 9, 9), "an.x0", 3.5, seed=1234), -3.0, 3.0)
    nz = h((6, 8, 9, 9), "an.noise", 2.0, seed=1234)
    t = torch.from_numpy(g["t"])
    sch = _fixture_scheduler(psg, golden)
    out = sch.add_noise(x0.to(DEV), nz.to(DEV), t.to(DEV)).cpu()
    assert np.array_equal(out.numpy(), g["out"]), "add_noise differs from the reference fixture bit-for-bit"
    # and against the oracle evaluated on THIS host (tables built here by the same torch ops)
    here = psg.NoiseScheduler()
    assert torch.equal(here.add_noise(x0.to(DEV), nz.to(DEV), t.to(DEV)).cpu(), O.add_noise(x0, nz, t, O.cosine_clipped_tables()))
    # fused clamp (trainer :363) on un-clamped input gives the same bits
    raw = h((6, 8, 9, 9), "an.x0", 3.5, seed=1234)
    out2 = sch.add_noise(raw.to(DEV), nz.to(DEV), t.to(DEV), clamp=True).cpu()
    assert np.array_equal(out2.numpy(), g["out"])
    # NaN/Inf fallback (:61-63), decided on the device
    bad = nz.clone()
    bad[2, 3, 4, 5] = float("inf")
    out3 = sch.add_noise(x0.to(DEV), bad.to(DEV), t.to(DEV)).cpu()
    assert np.array_equal(out3.numpy(), g["out_fallback"])
    # inf in the noise: the fallback x0 + 0.1*noise is inf too -> "fallback taken" (16) AND "still non-finite" (1): the
    # reference returns exactly these values (fixture) and the trainer's re-check then skips the batch (:376)
    assert int(sch.nan_flag(DEV).item()) == 17
    # a genuine rescue: a*x0 + b*noise overflows fp32, x0 + 0.1*noise does not -> bit 16 only, the batch TRAINS (:61-63)
    big = torch.full((1, 8, 3, 3), 3.0e38)
    tt = torch.tensor([500])
    got = sch.add_noise(big.to(DEV), big.to(DEV), tt.to(DEV)).cpu()
    assert int(sch.nan_flag(DEV).item()) == 16
    assert torch.equal(got, O.add_noise(big, big, tt, {k[4:]: torch.from_numpy(g2) for k, g2 in golden("schedule.npz").items() if k.startswith("cos_")}))
    assert bool(torch.isfinite(got).all())


def test_noise_add_edge_cases(psg):
    sch = psg.NoiseScheduler()
    # empty batch
    e = sch.add_noise(torch.zeros(0, 8, 27, 27, device=DEV), torch.zeros(0, 8, 27, 27, device=DEV), torch.zeros(0, dtype=torch.long, device=DEV))
    assert e.shape == (0, 8, 27, 27)
    # out-of-range timestep is flagged (bit 1), not a fault
    x = torch.ones(2, 8, 3, 3, device=DEV)
    sch.add_noise(x, x, torch.tensor([5, 1000], device=DEV))
    assert int(sch.nan_flag(DEV).item()) & 2
    # full-size batch, every timestep value: lookup exactness at B=1000
    t = torch.arange(1000)
    x0, nz = h((1000, 8, 4, 4), "big.x0", 2.0), h((1000, 8, 4, 4), "big.nz", 1.0)
    got = sch.add_noise(x0.to(DEV), nz.to(DEV), t.to(DEV)).cpu()
    assert torch.equal(got, O.add_noise(x0, nz, t, O.cosine_clipped_tables()))


def test_c_oracle_agrees(psg):
    """The plain-C oracle (oracle/noise_oracle.c) and the HIP kernel give identical bits."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle")], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(root, "oracle", "libpsg_oracle.so"))
    x0, nz = h((4, 8, 27, 27), "c.x0", 3.5), h((4, 8, 27, 27), "c.nz", 1.0)
    x0[3, 0, 0, :3] = torch.tensor([float("nan"), float("inf"), -float("inf")])      # clamp keeps NaN, bounds the infinities
    t = torch.tensor([0, 500, 999, 37], dtype=torch.int64)
    tb = O.cosine_clipped_tables()
    out = torch.empty_like(x0)
    f32p = lambda a: a.numpy().ctypes.data_as(C.c_void_p)
    lib.oracle_noise_add_f32(f32p(x0), f32p(nz), t.numpy().ctypes.data_as(C.c_void_p), f32p(tb["sqrt_alphas_cumprod"]),
                             f32p(tb["sqrt_one_minus_alphas_cumprod"]), f32p(out), None, C.c_int64(4), C.c_int64(8 * 27 * 27), 1)
    sch = psg.NoiseScheduler()
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    from pokemon_sprite_generator_amd import _lib
    got = torch.empty_like(x0, device=DEV)
    xd, nd, td = x0.to(DEV), nz.to(DEV), t.to(DEV)              # (kept alive: raw pointers are handed to the C ABI)
    _lib.check(_lib.init(0).psg_noise_add_f32(_lib.ptr(xd), _lib.ptr(nd), _lib.ptr(td),
                                              _lib.ptr(sch.to(DEV).sqrt_alphas_cumprod), _lib.ptr(sch.sqrt_one_minus_alphas_cumprod),
                                              _lib.ptr(got), _lib.ptr(flag), 4, 8 * 27 * 27, 1000, 1, _lib.stream_ptr()), "noise_add")
    torch.cuda.synchronize()
    got = got.cpu()
    ref = torch.clamp(x0, -3.0, 3.0)                                                  # torch's own clamp semantics
    assert torch.isnan(out[3, 0, 0, 0]) and torch.isnan(got[3, 0, 0, 0]) and torch.isnan(ref[3, 0, 0, 0])
    assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(out, nan=7.0))
    assert int(flag.item()) == 16                                                     # the NaN asks for the fallback


# ---------------------------------------------------------------- a-8 SmoothL1 / a-9 ddpm update
def test_smooth_l1(psg):
    from pokemon_sprite_generator_amd import _lib
    lib = _lib.init(0)
    pred, tgt = h((4, 8, 27, 27), "sl.p", 1.0), h((4, 8, 27, 27), "sl.t", 1.0)
    pred[0, 0, 0, :8] = tgt[0, 0, 0, :8] + torch.tensor([0.05, -0.05, 0.1, -0.1, 0.0999, 0.2, -3.0, 0.0])   # both branches + boundary
    p = pred.clone().requires_grad_(True)
    ref = O.smooth_l1(p, tgt, 0.1)
    ref.backward()
    pd, td = pred.to(DEV), tgt.to(DEV)
    grad, loss = torch.empty_like(pd), torch.zeros(1, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(lib.psg_reduce_workspace_bytes(), dtype=torch.uint8, device=DEV)
    _lib.check(lib.psg_smooth_l1_f32(_lib.ptr(pd), _lib.ptr(td), _lib.ptr(grad), _lib.ptr(loss), _lib.ptr(flag), 0.1, 1.0, pd.numel(),
                                     _lib.ptr(ws), _lib.stream_ptr()))
    assert abs(float(loss.item()) - float(ref)) / float(ref) < 1e-5          # fp32 reduction order only
    assert maxrel(grad.cpu(), p.grad) < 1e-6
    assert torch.equal(F.smooth_l1_loss(pred, tgt, beta=0.1), ref.detach()) or abs(float(F.smooth_l1_loss(pred, tgt, beta=0.1)) - float(ref)) < 1e-6
    assert int(flag.item()) == 0


def test_ddpm_update_bit_exact(psg):
    from pokemon_sprite_generator_amd import _lib
    lib = _lib.init(0)
    sch = psg.NoiseScheduler()
    tb = O.cosine_clipped_tables()
    c1, c2, sg = sch.step_tables(DEV)
    for t in (999, 950, 500, 50, 1, 0):
        x, eps, z = h((3, 8, 27, 27), f"dd.x{t}", 2.0), h((3, 8, 27, 27), f"dd.e{t}", 1.0), h((3, 8, 27, 27), f"dd.z{t}", 1.0)
        a, b, s = O.ddpm_step_coeffs(tb, t)
        ref = a * (x - b * eps)
        if t > 0:
            ref = ref + s * z
        xd, ed, zd = x.to(DEV).clone(), eps.to(DEV), z.to(DEV)
        td = torch.tensor([t], dtype=torch.int32, device=DEV)
        _lib.check(lib.psg_ddpm_update_f32(_lib.ptr(xd), _lib.ptr(ed), _lib.ptr(zd), _lib.ptr(c1), _lib.ptr(c2), _lib.ptr(sg),
                                           _lib.ptr(td), xd.numel(), _lib.stream_ptr()))
        assert torch.equal(xd.cpu(), ref), f"ddpm update differs at t={t}"


# ---------------------------------------------------------------- small ops
@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_pool_sinusoid(psg, dtype):
    from pokemon_sprite_generator_amd import ops
    x = h((3, 8, 27, 27), "lay.x", 2.0)
    cl = ops.nchw_to_nhwc(x.to(DEV), dtype)
    assert cl.shape == (3, 27, 27, 8)
    assert torch.equal(cl.float().cpu(), to_cl(x, dtype).float())
    back = ops.nhwc_to_nchw(cl)
    assert torch.equal(back.cpu(), _q(x, dtype))
    text = h((3, 32, 256), "lay.t", 1.5)
    pooled, cast = ops.text_pool(text.to(DEV), dtype)
    assert maxrel(pooled.float().cpu(), text.mean(dim=1)) < TOL[dtype] * 0.5
    assert torch.equal(cast.float().cpu(), _q(text, dtype))
    t = torch.tensor([0, 1, 500, 999, 37, 250])
    coeff = torch.exp(torch.arange(64) * -(math.log(10000) / 63))
    e = t.float().unsqueeze(-1) * coeff.unsqueeze(0)
    ref = torch.cat([torch.sin(e), torch.cos(e)], dim=-1)
    got = ops.timestep_sinusoid(t.to(DEV), coeff.to(DEV), dtype).float().cpu()
    assert (got - ref).abs().max() < (2e-6 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hi,ho", [(4, 7), (7, 14), (14, 27)])
def test_upsample(psg, dtype, hi, ho):
    from pokemon_sprite_generator_amd import ops
    x = _q(h((2, 64, hi, hi), f"up.x{hi}", 1.5), dtype).requires_grad_(True)
    g = _q(h((2, 64, ho, ho), f"up.g{hi}", 1.0), dtype)
    ref = F.interpolate(x, size=(ho, ho), mode="bilinear", align_corners=False)
    ref.backward(g)
    xd = to_cl(x.detach(), dtype).to(DEV).requires_grad_(True)
    y = ops.upsample_bilinear(xd, (ho, ho))
    y.backward(to_cl(g, dtype).to(DEV))
    assert maxrel(from_cl(y.cpu()), ref) < TOL[dtype]
    assert maxrel(from_cl(xd.grad.cpu()), x.grad) < TOL[dtype]


# ---------------------------------------------------------------- GroupNorm (+SiLU)
GN_CASES = [  # B, HW (as H), C, eps, silu
    (2, 7, 64, 1e-5, True), (2, 27, 320, 1e-5, True), (1, 14, 1280, 1e-5, True), (2, 4, 2560, 1e-5, True),
    (2, 27, 640, 1e-5, True), (3, 7, 1280, 1e-6, False), (2, 5, 128, 1e-6, False),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,C,eps,silu", GN_CASES)
def test_groupnorm(psg, dtype, B, H, C, eps, silu):
    from pokemon_sprite_generator_amd import ops
    G = 32
    x = _q(h((B, C, H, H), f"gn.x{C}", 1.3) + 0.4, dtype).requires_grad_(True)
    gamma = (1.0 + h((C,), f"gn.g{C}", 0.3)).requires_grad_(True)
    beta = h((C,), f"gn.b{C}", 0.2).requires_grad_(True)
    gy = _q(h((B, C, H, H), f"gn.gy{C}", 1.0), dtype)
    ref = F.group_norm(x, G, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    ref.backward(gy)
    xd = to_cl(x.detach(), dtype).to(DEV).requires_grad_(True)
    gd, bd = gamma.detach().to(DEV).requires_grad_(True), beta.detach().to(DEV).requires_grad_(True)
    y = ops.group_norm(xd, gd, bd, G, eps, silu)
    y.backward(to_cl(gy, dtype).to(DEV))
    tol = TOL[dtype]
    assert maxrel(from_cl(y.cpu()), ref) < tol
    assert maxrel(from_cl(xd.grad.cpu()), x.grad) < tol * 2
    assert maxrel(gd.grad.cpu(), gamma.grad) < tol * 2
    assert maxrel(bd.grad.cpu(), beta.grad) < tol * 2


# ---------------------------------------------------------------- conv / linear
CONV_CASES = [  # B, H, Cin, Cout, ks, stride
    (2, 7, 64, 128, 3, 1), (3, 5, 64, 64, 3, 1), (2, 9, 64, 64, 3, 2), (2, 14, 128, 64, 3, 2), (2, 27, 64, 64, 3, 2),
    (2, 4, 128, 64, 1, 1), (2, 27, 8, 320, 3, 1), (2, 27, 320, 8, 3, 1), (1, 4, 2560, 128, 3, 1), (5, 7, 192, 320, 3, 1),
    (1, 27, 320, 320, 3, 1), (2, 14, 64, 320, 1, 1), (3, 7, 96, 160, 3, 1),      # (Cout 320 / 160: the weight gradient's 160-row tiles)
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Cin,Cout,ks,stride", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(psg, dtype, B, H, Cin, Cout, ks, stride):
    from pokemon_sprite_generator_amd import ops
    pad = 1 if ks == 3 else 0
    name = f"cv{B}.{H}.{Cin}.{Cout}.{ks}.{stride}"
    x = _q(h((B, Cin, H, H), name + "x", 1.2), dtype).requires_grad_(True)
    w = _q(h((Cout, Cin, ks, ks), name + "w", math.sqrt(3.0 / (Cin * ks * ks))), dtype).requires_grad_(True)
    b = h((Cout,), name + "b", 0.2).requires_grad_(True)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    Ho = ref.shape[-1]
    ra = _q(h((B, Cout), name + "ra", 0.5), dtype).requires_grad_(True)
    res = _q(h((B, Cout, Ho, Ho), name + "res", 1.0), dtype).requires_grad_(True)
    ref = ref + ra[:, :, None, None] + res
    gy = _q(h(tuple(ref.shape), name + "gy", 1.0), dtype)
    ref.backward(gy)
    xd = to_cl(x.detach(), dtype).to(DEV).requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    bd = b.detach().to(DEV).requires_grad_(True)
    rad = ra.detach().to(dtype).to(DEV).requires_grad_(True)
    resd = to_cl(res.detach(), dtype).to(DEV).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, stride=stride, rowadd=rad, residual=resd)
    y.backward(to_cl(gy, dtype).to(DEV))
    tol = TOL[dtype]
    assert maxrel(from_cl(y.cpu()), ref) < tol, "forward"
    assert maxrel(from_cl(xd.grad.cpu()), x.grad) < tol, "dgrad"
    assert maxrel(wd.grad.cpu(), w.grad) < tol, "wgrad"
    assert maxrel(bd.grad.cpu(), b.grad) < tol, "dbias"
    assert maxrel(rad.grad.float().cpu(), ra.grad) < tol, "d rowadd"
    assert maxrel(from_cl(resd.grad.cpu()), res.grad) < tol, "d residual"


@pytest.mark.parametrize("B,H,Cin,Cout", [(256, 14, 64, 128), (512, 7, 128, 128), (72, 27, 64, 256), (2048, 4, 64, 128),
                                          (256, 14, 64, 320), (72, 27, 128, 320)])
def test_conv3x3_large_tiles(psg, B, H, Cin, Cout):
    """Shapes large enough for the tile chooser to pick the 128x128 / 128x160 tiles the benchmark runs on (the small
    parity cases above mostly take 64x64): bf16 forward, data gradient AND weight / bias gradient against a CPU fp32
    convolution of the same bf16 operands, for every map width of the U-Net (27, 14, 7, 4).  The Cout = 320 cases take the
    160-row weight-gradient tile at M = 50 176 / 52 488 pixels with a multi-slab split-K (the benchmark's regime)."""
    from pokemon_sprite_generator_amd import ops
    dtype = torch.bfloat16
    name = f"halo{B}.{H}.{Cin}.{Cout}"
    x = _q(h((B, Cin, H, H), name + "x", 1.2), dtype).requires_grad_(True)
    w = _q(h((Cout, Cin, 3, 3), name + "w", math.sqrt(3.0 / (Cin * 9))), dtype).requires_grad_(True)
    b = h((Cout,), name + "b", 0.2).requires_grad_(True)
    ref = F.conv2d(x, w, b, padding=1)
    gy = _q(h(tuple(ref.shape), name + "gy", 1.0), dtype)
    ref.backward(gy)
    xd = to_cl(x.detach(), dtype).to(DEV).requires_grad_(True)
    for layout in ("oihw", "ohwi"):                 # torch-contiguous weights, and the parameter arena's channels_last order
        wd = w.detach().to(DEV)
        if layout == "ohwi":
            wd = wd.contiguous(memory_format=torch.channels_last)
        wd = torch.nn.Parameter(wd)
        bd = torch.nn.Parameter(b.detach().to(DEV))
        xd.grad = None
        y = ops.conv2d(xd, wd, bd)
        y.backward(to_cl(gy, dtype).to(DEV))
        tol = TOL[dtype]
        assert maxrel(from_cl(y.cpu()), ref.detach()) < tol, "forward"
        assert maxrel(from_cl(xd.grad.cpu()), x.grad) < tol, "dgrad"
        # fp32 accumulation of exact bf16 products: only the summation order differs from the CPU's
        assert maxrel(wd.grad.cpu(), w.grad) < 1e-4, f"wgrad ({layout})"
        assert rel_l2(wd.grad.cpu(), w.grad) < 1e-5, f"wgrad rel-L2 ({layout})"
        assert maxrel(bd.grad.cpu(), b.grad) < 1e-4, f"bias gradient ({layout})"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Cin,Cout,stride,uses", [(2, 9, 64, 96, 1, 1), (3, 14, 160, 320, 2, 1), (40, 27, 32, 64, 1, 2)])
def test_conv_ohwi_master_weights(psg, dtype, B, H, Cin, Cout, stride, uses):
    """Conv weights stored channels_last (OHWI, optim.ParamArena's order): prepared weights and the weight
    gradient (direct single-split write, split-K sum, and the accumulate path of a weight used twice through a
    GradSink) must equal the OIHW results; the gradient comes back in the parameter's own memory order."""
    from pokemon_sprite_generator_amd import ops, GradArena
    name = f"ohwi{B}.{H}.{Cin}.{Cout}.{stride}"
    x = _q(h((B, Cin, H, H), name + "x", 1.2), dtype).requires_grad_(True)
    w = _q(h((Cout, Cin, 3, 3), name + "w", math.sqrt(3.0 / (Cin * 9))), dtype).requires_grad_(True)
    ref = F.conv2d(x, w, None, stride=stride, padding=1)
    gy = _q(h(tuple(ref.shape), name + "gy", 1.0), dtype)
    (ref * float(uses)).backward(gy)
    xd = to_cl(x.detach(), dtype).to(DEV).requires_grad_(True)
    wd = torch.nn.Parameter(w.detach().to(DEV).contiguous(memory_format=torch.channels_last))
    assert ops.weight_layout(wd) == ops.W_OHWI
    arena = GradArena([wd]) if uses > 1 else None
    if arena is not None:
        arena.zero()
    y = ops.conv2d(xd, wd, None, stride=stride)
    for _ in range(uses - 1):
        y = y + ops.conv2d(xd, wd, None, stride=stride)
    y.backward(to_cl(gy, dtype).to(DEV))
    tol = TOL[dtype]
    assert maxrel(from_cl(y.cpu()), ref * float(uses)) < tol, "forward"
    assert maxrel(from_cl(xd.grad.cpu()), x.grad) < tol, "dgrad"
    assert wd.grad.stride() == wd.stride(), "gradient must share the parameter's memory order"
    assert maxrel(wd.grad.cpu(), w.grad) < tol, "wgrad"
    if arena is not None:
        arena.release()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", ["none", "silu", "gelu"])
def test_linear_epilogue(psg, dtype, act):
    from pokemon_sprite_generator_amd import ops
    M, K, N = 150, 256, 192
    x = _q(h((3, 50, K), "li.x" + act, 1.0), dtype).requires_grad_(True)
    w = _q(h((N, K), "li.w" + act, math.sqrt(3.0 / K)), dtype).requires_grad_(True)
    b = h((N,), "li.b" + act, 0.3).requires_grad_(True)
    res = _q(h((3, 50, N), "li.r" + act, 1.0), dtype).requires_grad_(True)
    u = F.linear(x, w, b)
    a = {"none": u, "silu": F.silu(u), "gelu": F.gelu(u)}[act]
    ref = res + 0.7 * a
    gy = _q(h((3, 50, N), "li.g" + act, 1.0), dtype)
    ref.backward(gy)
    code = {"none": ops.ACT_NONE, "silu": ops.ACT_SILU, "gelu": ops.ACT_GELU}[act]
    xd, wd, bd = x.detach().to(dtype).to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    rd = res.detach().to(dtype).to(DEV).requires_grad_(True)
    y = ops.linear(xd, wd, bd, residual=rd, act=code, alpha=0.7)
    y.backward(gy.to(dtype).to(DEV))
    tol = TOL[dtype]
    assert maxrel(y.float().cpu(), ref) < tol
    assert maxrel(xd.grad.float().cpu(), x.grad) < tol * 2
    assert maxrel(wd.grad.cpu(), w.grad) < tol * 2
    assert maxrel(bd.grad.cpu(), b.grad) < tol * 2
    assert maxrel(rd.grad.float().cpu(), res.grad) < tol


def test_linear_dropout_consistent(psg):
    """Dropout mask: right keep-rate, and the SAME mask in forward and backward."""
    from pokemon_sprite_generator_amd import ops
    p = 0.3
    x = h((64, 128), "dr.x", 1.0).to(DEV).requires_grad_(True)
    w = h((256, 128), "dr.w", 0.2).to(DEV).requires_grad_(True)
    y = ops.linear(x, w, None, drop_p=p, seed=77)
    keep = (y != 0).float()
    rate = float(keep.mean())
    assert abs(rate - (1 - p)) < 0.02, rate
    ref = (x.detach() @ w.detach().t()) * keep / (1 - p)
    assert maxrel(y, ref) < 1e-4
    g = h((64, 256), "dr.g", 1.0).to(DEV)
    y.backward(g)
    gm = g * keep / (1 - p)
    assert maxrel(x.grad, gm @ w.detach()) < 1e-4
    assert maxrel(w.grad, gm.t() @ x.detach()) < 1e-4
    y2 = ops.linear(x.detach(), w.detach(), None, drop_p=p, seed=78)
    assert not torch.equal((y2 != 0), (y != 0))


def test_conv_linearity_full_size(psg):
    """Size-independent property at the benchmark shape (B=256, 27x27x320): conv(a*x1 + x2) == a*conv(x1) + conv(x2) (bf16 path)."""
    from pokemon_sprite_generator_amd import ops
    torch.manual_seed(0)
    x1 = torch.randn(256, 27, 27, 320, device=DEV).bfloat16()
    x2 = torch.randn(256, 27, 27, 320, device=DEV).bfloat16()
    w = torch.randn(320, 320, 3, 3, device=DEV) * 0.02
    with torch.no_grad():
        y1, y2 = ops.conv2d(x1, w), ops.conv2d(x2, w)
        xs = (2.0 * x1.float() + x2.float()).bfloat16()
        ys = ops.conv2d(xs, w)
    err = rel_l2(ys.float(), 2.0 * y1.float() + y2.float())
    assert err < 2e-2, err
    # spot-check 64 random output pixels against a CPU fp32 conv of the same bf16 inputs
    idx = torch.randint(0, 256, (4,)).tolist()
    ref = F.conv2d(x1[idx].float().cpu().permute(0, 3, 1, 2), w.bfloat16().float().cpu(), padding=1)
    assert maxrel(from_cl(y1[idx].cpu()), ref) < 2e-2


# ---------------------------------------------------------------- persistent pointwise kernel (csrc/conv_pw.hip)
def _pw(on):
    from pokemon_sprite_generator_amd import _lib
    _lib.check(_lib.init(0).psg_conv_set_pw(int(on)), "psg_conv_set_pw")


def _pw_count():
    from pokemon_sprite_generator_amd import _lib
    return int(_lib.init(0).psg_conv_pw_launches())


@pytest.mark.parametrize("M,K,N", [(12800, 256, 1024), (50176, 640, 640), (12544, 1280, 1280)])
def test_pointwise_persistent_kernel_is_bitwise_the_tiled_kernel(psg, M, K, N):
    """conv_pw_kernel (persistent workgroups, next tile's K slices requested before the epilogue, strip-staged stores, counted
    vmcnt waits) against conv_gemm_kernel on the same launches: every epilogue kind the U-Net's Linears use - bias, gate +
    residual, dropout + residual, GELU + dropout + saved derivative (two outputs), the derivative-multiplying data gradient,
    plain data gradients - forward AND backward results must be identical BIT FOR BIT (same MFMA order, same epilogue
    arithmetic), and the persistent kernel must really have taken the launches.  Reference check of the same ops: test_linear_epilogue,
    test_ffn_gelu_dropout_backward (small M, tiled kernel) - so equality here carries their parity over."""
    from pokemon_sprite_generator_amd import ops
    dt = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(M + K)
    x = torch.randn(M, K, device=DEV, generator=g).to(dt)
    res = torch.randn(M, N, device=DEV, generator=g).to(dt)
    w = (torch.randn(N, K, device=DEV, generator=g) * math.sqrt(1.0 / K))
    b = torch.randn(N, device=DEV, generator=g) * 0.3
    gy = torch.randn(M, N, device=DEV, generator=g).to(dt)
    w1 = torch.randn(2 * K, K, device=DEV, generator=g) * math.sqrt(1.0 / K)
    b1 = torch.randn(2 * K, device=DEV, generator=g) * 0.3
    w2 = torch.randn(K, 2 * K, device=DEV, generator=g) * math.sqrt(0.5 / K)
    b2 = torch.randn(K, device=DEV, generator=g) * 0.3
    gx = torch.randn(M, K, device=DEV, generator=g).to(dt)

    def run():
        outs = []
        xs = x.clone().requires_grad_(True)
        rs = res.clone().requires_grad_(True)
        y = ops.linear(xs, w.clone().requires_grad_(True), b.clone().requires_grad_(True))                     # bias only
        y.backward(gy)
        outs += [y.detach(), xs.grad.clone()]
        xs.grad = None
        ws, bs = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.linear(xs, ws, bs, residual=rs, alpha=0.7)                                                     # out-proj form
        y.backward(gy)
        outs += [y.detach(), xs.grad.clone(), rs.grad.clone(), ws.grad.clone(), bs.grad.clone()]
        xs.grad = None
        y = ops.linear(xs, w.clone().requires_grad_(True), b.clone().requires_grad_(True), residual=rs, alpha=0.6, drop_p=0.05, seed=991)
        y.backward(gy)
        outs += [y.detach(), xs.grad.clone()]
        xs.grad = None
        y = ops.linear(xs, w.clone().requires_grad_(True), b.clone().requires_grad_(True), act=ops.ACT_GELU)
        y.backward(gy)
        outs += [y.detach(), xs.grad.clone()]
        xs.grad = None
        p1 = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
        y = ops.ffn(xs, p1[0], p1[1], p1[2], p1[3], 0.6, drop_p=0.05, seed1=17, seed2=18)                      # fused FFN node (train mode)
        y.backward(gx)
        outs += [y.detach(), xs.grad.clone()] + [t.grad.clone() for t in p1]
        torch.cuda.synchronize()
        return outs

    try:
        _pw(0)
        c0 = _pw_count()
        ref = run()
        assert _pw_count() == c0
        _pw(1)
        got = run()
        # (a launch qualifies with >= 768 output tiles: at the small shape only the four plain forwards do, at 14x14x640 the
        #  forwards, the data gradients and all four GEMMs of the FFN node)
        assert _pw_count() - c0 >= {12800: 4, 12544: 12, 50176: 12}[M], _pw_count() - c0
    finally:
        _pw(1)
    for i, (a, r) in enumerate(zip(got, ref)):
        assert torch.isfinite(r.float()).all()
        assert torch.equal(a, r), (i, float((a.float() - r.float()).abs().max()))


# ---------------------------------------------------------------- attention core
ATTN_CASES = [  # B, heads, L, S, d, self
    (2, 8, 16, 16, 16, True), (2, 4, 49, 49, 32, True), (1, 8, 196, 196, 80, True), (2, 8, 16, 32, 160, False),
    (2, 8, 196, 32, 80, False), (1, 4, 49, 20, 320, False), (1, 8, 49, 32, 160, False),
    # the reference CLI's real head count: 4 heads at 1280 channels -> head_dim 320 (improved_diffusion_trainer.py:215)
    (2, 4, 49, 49, 320, True), (2, 4, 16, 16, 320, True), (2, 4, 16, 32, 320, False), (1, 4, 49, 32, 320, False),
    (2, 8, 40, 20, 64, False),                       # head_dim 64: the VAE decoder's first cross-attention (512 channels, 8 heads)
]


def _attn_ref(q, k, v, heads):
    B, L, E = q.shape
    S, d = k.shape[1], E // heads
    qh = q.view(B, L, heads, d).transpose(1, 2) * math.sqrt(1.0 / d)
    kh = k.view(B, S, heads, d).transpose(1, 2)
    vh = v.view(B, S, heads, d).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-2, -1), dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, L, E)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,heads,L,S,d,self_mode", ATTN_CASES)
def test_attention(psg, dtype, B, heads, L, S, d, self_mode):
    from pokemon_sprite_generator_amd import ops
    E = heads * d
    name = f"at{B}.{heads}.{L}.{S}.{d}"
    if self_mode:
        qkv = _q(h((B, L, 3 * E), name, 1.0), dtype).requires_grad_(True)
        ref = _attn_ref(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], heads)
    else:
        qs = _q(h((B, L, E), name + "q", 1.0), dtype).requires_grad_(True)
        kv = _q(h((B, S, 2 * E), name + "kv", 1.0), dtype).requires_grad_(True)
        ref = _attn_ref(qs, kv[..., :E], kv[..., E:], heads)
    go = _q(h((B, L, E), name + "g", 1.0), dtype)
    ref.backward(go)
    tol = TOL[dtype]
    if self_mode:
        qd = qkv.detach().to(dtype).to(DEV).requires_grad_(True)
        o = ops.attention_self(qd, heads)
        o.backward(go.to(dtype).to(DEV))
        assert maxrel(o.float().cpu(), ref) < tol
        assert maxrel(qd.grad.float().cpu(), qkv.grad) < tol * 2
    else:
        qd = qs.detach().to(dtype).to(DEV).requires_grad_(True)
        kd = kv.detach().to(dtype).to(DEV).requires_grad_(True)
        o = ops.attention_cross(qd, kd, heads)
        o.backward(go.to(dtype).to(DEV))
        assert maxrel(o.float().cpu(), ref) < tol
        assert maxrel(qd.grad.float().cpu(), qs.grad) < tol * 2
        assert maxrel(kd.grad.float().cpu(), kv.grad) < tol * 2


def _attn_paths():
    from pokemon_sprite_generator_amd import _lib
    a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.init(0).psg_attn_path_counts(C.byref(a), C.byref(b), C.byref(c)), "psg_attn_path_counts")
    return a.value, b.value, c.value


@pytest.mark.parametrize("d", [16, 32, 80, 160, 320])
def test_attention_bf16_takes_the_mfma_kernels(psg, d):
    """bf16 attention at every head_dim the U-Net produces (8 heads: 80 / 160; the CLI's 4 heads: 160 / 320) must run on
    the matrix cores, forward and backward; so must fp32 (exact-fp32 MFMA) up to head_dim 160 - head_dim 320 in fp32 does
    not fit K and V in LDS at 7x7 and runs the VALU kernels."""
    from pokemon_sprite_generator_amd import ops
    heads, L = 4, 49
    qkv = h((2, L, 3 * heads * d), f"path{d}", 1.0)
    for dtype, col in ((torch.bfloat16, 0), (torch.float32, 2 if d <= 160 else 1)):
        before = _attn_paths()
        x = qkv.to(dtype).to(DEV).requires_grad_(True)
        ops.attention_self(x, heads).sum().backward()
        after = _attn_paths()
        for c in range(3):
            assert after[c] - before[c] == (2 if c == col else 0), (dtype, before, after)


@pytest.mark.parametrize("S", [49, 20, 32])
def test_attention_dropout_mask_is_path_independent(psg, S):
    """The VALU (fp32) and MFMA (bf16) kernels regenerate the SAME stateless dropout mask from (seed, row, key) - for odd
    S too (7x7 self-attention: S = 49) - so a forward on one path can be differentiated on the other.  With q = k = 0 and
    v = key one-hot columns, output column j of a query row is (1/S)/(1-p) where key j was kept and 0 where it was dropped."""
    from pokemon_sprite_generator_amd import ops
    B, heads, L, d, p = 2, 2, 49, 80, 0.3
    E = heads * d
    q = torch.zeros(B, L, E)
    kv = torch.zeros(B, S, 2 * E)
    for hh in range(heads):
        for j in range(S):
            kv[:, j, E + hh * d + j] = 1.0           # v[key j] = e_j inside each head
    o32 = ops.attention_cross(q.to(DEV), kv.to(DEV), heads, p, 987).cpu()
    o16 = ops.attention_cross(q.bfloat16().to(DEV), kv.bfloat16().to(DEV), heads, p, 987).float().cpu()
    keep32 = o32.view(B, L, heads, d)[..., :S] > 0
    keep16 = o16.view(B, L, heads, d)[..., :S] > 0
    assert 0.55 < float(keep32.float().mean()) < 0.85            # p = 0.3 really drops
    assert torch.equal(keep32, keep16)


def test_attention_softmax_spike(psg):
    """A forced large logit (one key aligned with one query) must not overflow: softmax is max-subtracted."""
    from pokemon_sprite_generator_amd import ops
    B, heads, L, d = 1, 2, 32, 16
    E = heads * d
    qkv = h((B, L, 3 * E), "spike", 1.0)
    qkv[0, 3, :E] = 60.0
    qkv[0, 5, E:2 * E] = 60.0
    ref = _attn_ref(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], heads)
    o = ops.attention_self(qkv.to(DEV), heads)
    assert torch.isfinite(o).all()
    assert maxrel(o.cpu(), ref) < 1e-3


def test_attention_dropout_statistics(psg):
    from pokemon_sprite_generator_amd import ops
    B, heads, L, d = 4, 8, 49, 16
    E = heads * d
    qkv = torch.zeros(B, L, 3 * E)
    qkv[..., 2 * E:] = 1.0                      # v = 1 -> o = sum_j drop(P)_j ; E[o] = 1, Var = p/((1-p) L)
    o = ops.attention_self(qkv.to(DEV), heads, 0.05, 123).cpu()
    assert abs(float(o.mean()) - 1.0) < 0.01
    assert 0.5 * 0.05 / (0.95 * L) < float(o.var()) < 2.0 * 0.05 / (0.95 * L)


# ---------------------------------------------------------------- train-mode (dropout ON) gradient parity
# The benchmarked step runs with p = 0.05 on the attention probabilities and twice in the FFN.  The masks are a stateless
# hash of (seed, element index) REGENERATED inside the backward kernels, so a wrong row base / key offset there gives
# plausible-looking wrong gradients that no self-comparison sees.  These tests read the forward's keep-mask off probe
# launches (the mask depends on seed and geometry only, never on the values), build the torch fp32 reference
#   O = (softmax(QK^T/sqrt(d)) * keep / (1-p)) V          (torch/nn/functional.py: dropout on the probabilities)
# with that mask and compare o, dq, dk, dv - per kernel family (bf16 MFMA, exact-fp32 MFMA, VALU via psg_attn_set_paths).
ATTN_DROP_CASES = [  # B, heads, L, S, d, self   (unet.py:160-173 at 8 heads: d 80/160; the CLI's 4 heads: d 320)
    (2, 8, 196, 196, 80, True), (2, 8, 196, 32, 80, False), (2, 8, 49, 49, 160, True), (2, 8, 49, 32, 160, False),
    (2, 8, 16, 16, 160, True), (2, 8, 16, 32, 160, False), (2, 4, 49, 49, 320, True), (1, 4, 49, 20, 320, False),
    (1, 4, 49, 21, 80, False),                       # odd S: the pair hash's last key has no partner
]


def _attn_set_paths(mask):
    from pokemon_sprite_generator_amd import _lib
    _lib.check(_lib.init(0).psg_attn_set_paths(int(mask)), "psg_attn_set_paths")


def _attn_keep_mask(ops, B, heads, L, S, d, self_mode, p, seed, dtype):
    """keep[b, h, l, s] of the forward's dropout mask: q = k = 0 -> uniform probabilities 1/S; v = key one-hot columns,
    d keys per probe launch -> output column j of a query row is (1/S)/(1-p) where key c0+j was kept, 0 where dropped."""
    E = heads * d
    keep = torch.zeros(B, heads, L, S, dtype=torch.bool)
    for c0 in range(0, S, d):
        n = min(d, S - c0)
        if self_mode:
            src = torch.zeros(B, L, 3 * E)
            vv = src[..., 2 * E:]
        else:
            q = torch.zeros(B, L, E)
            src = torch.zeros(B, S, 2 * E)
            vv = src[..., E:]
        for hh in range(heads):
            for j in range(n):
                vv[:, c0 + j, hh * d + j] = 1.0
        with torch.no_grad():
            if self_mode:
                o = ops.attention_self(src.to(dtype).to(DEV), heads, p, seed)
            else:
                o = ops.attention_cross(q.to(dtype).to(DEV), src.to(dtype).to(DEV), heads, p, seed)
        keep[..., c0:c0 + n] = (o.float().cpu().view(B, L, heads, d)[..., :n] > 0).permute(0, 2, 1, 3)
    return keep


def _attn_ref_drop(q, k, v, heads, keep, p):
    B, L, E = q.shape
    S, d = k.shape[1], E // heads
    qh = q.view(B, L, heads, d).transpose(1, 2) * math.sqrt(1.0 / d)
    kh = k.view(B, S, heads, d).transpose(1, 2)
    vh = v.view(B, S, heads, d).transpose(1, 2)
    pr = torch.softmax(qh @ kh.transpose(-2, -1), dim=-1) * keep.to(q.dtype) / (1.0 - p)
    return (pr @ vh).transpose(1, 2).reshape(B, L, E)


@pytest.mark.parametrize("path", ["mfma_bf16", "mfma_f32", "valu_f32", "valu_bf16"])
@pytest.mark.parametrize("B,heads,L,S,d,self_mode", ATTN_DROP_CASES)
def test_attention_backward_with_dropout(psg, path, B, heads, L, S, d, self_mode):
    from pokemon_sprite_generator_amd import ops
    dtype = torch.bfloat16 if path.endswith("bf16") else torch.float32
    col = {"mfma_bf16": 0, "valu_f32": 1, "valu_bf16": 1, "mfma_f32": 2}[path]
    if path == "mfma_f32" and d > 160:
        pytest.skip("fp32 head_dim 320 has no exact-fp32 MFMA kernel (K, V do not fit LDS): the VALU case covers it")
    p, seed = 0.3, 24680 + 7 * L + S
    E = heads * d
    name = f"atd{B}.{heads}.{L}.{S}.{d}"
    _attn_set_paths(0 if path.startswith("valu") else 3)
    try:
        keep = _attn_keep_mask(ops, B, heads, L, S, d, self_mode, p, seed, dtype)
        rate = float(keep.float().mean())
        assert abs(rate - (1 - p)) < 0.03, rate
        if self_mode:
            qkv = _q(h((B, L, 3 * E), name, 1.0), dtype).requires_grad_(True)
            ref = _attn_ref_drop(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], heads, keep, p)
        else:
            qs = _q(h((B, L, E), name + "q", 1.0), dtype).requires_grad_(True)
            kv = _q(h((B, S, 2 * E), name + "kv", 1.0), dtype).requires_grad_(True)
            ref = _attn_ref_drop(qs, kv[..., :E], kv[..., E:], heads, keep, p)
        go = _q(h((B, L, E), name + "g", 1.0), dtype)
        ref.backward(go)
        before = _attn_paths()
        if self_mode:
            qd = qkv.detach().to(dtype).to(DEV).requires_grad_(True)
            o = ops.attention_self(qd, heads, p, seed)
            o.backward(go.to(dtype).to(DEV))
            got = [("o", o, ref), ("dqkv", qd.grad, qkv.grad), ("dq", qd.grad[..., :E], qkv.grad[..., :E]),
                   ("dk", qd.grad[..., E:2 * E], qkv.grad[..., E:2 * E]), ("dv", qd.grad[..., 2 * E:], qkv.grad[..., 2 * E:])]
        else:
            qd = qs.detach().to(dtype).to(DEV).requires_grad_(True)
            kd = kv.detach().to(dtype).to(DEV).requires_grad_(True)
            o = ops.attention_cross(qd, kd, heads, p, seed)
            o.backward(go.to(dtype).to(DEV))
            got = [("o", o, ref), ("dq", qd.grad, qs.grad), ("dk", kd.grad[..., :E], kv.grad[..., :E]), ("dv", kd.grad[..., E:], kv.grad[..., E:])]
        after = _attn_paths()
        for c in range(3):                            # forward + backward both ran on the family under test
            assert after[c] - before[c] == (2 if c == col else 0), (path, before, after)
        tol = TOL[dtype]                              # the bars of test_attention (dropout off)
        for what, a, b in got:
            assert maxrel(a.float().cpu(), b) < (tol if what == "o" else tol * 2), (path, what, maxrel(a.float().cpu(), b))
    finally:
        _attn_set_paths(3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C", [(2 * 196, 640), (300, 256)])
def test_ffn_gelu_dropout_backward(psg, dtype, M, C):
    """The fused FFN node in TRAIN mode (unet.py:176-187,250): y = x + 0.6 * drop2(W2 drop1(gelu(W1 x + b1)) + b2) with p = 0.3.
    Covers the GELU + dropout + saved-derivative epilogue (EK_GELU_DROP, PSG_CONV_SAVE_DACT / DACT_MUL) and the second
    dropout's regenerated mask (psg_epilogue_bwd) against torch autograd with the masks read off probe launches."""
    from pokemon_sprite_generator_amd import ops
    p, s1, s2, alpha = 0.3, 1357, 2468, 0.6
    Hd = 2 * C
    nm = f"ffd{M}.{C}"
    # probes: the mask is a function of (seed, element index) only
    with torch.no_grad():
        ones1 = torch.ones(Hd, device=DEV)
        z1 = ops.linear(torch.zeros(M, C, device=DEV, dtype=dtype), torch.zeros(Hd, C, device=DEV), ones1, act=ops.ACT_GELU, drop_p=p, seed=s1)
        m1 = (z1.float().cpu() > 0)
        z2 = ops.linear(torch.zeros(M, Hd, device=DEV, dtype=dtype), torch.zeros(C, Hd, device=DEV), torch.ones(C, device=DEV), drop_p=p, seed=s2)
        m2 = (z2.float().cpu() > 0)
    assert abs(float(m1.float().mean()) - (1 - p)) < 0.02 and abs(float(m2.float().mean()) - (1 - p)) < 0.02
    assert not torch.equal(m1[:, :C], m2)
    x = _q(h((M, C), nm + "x", 1.0), dtype).requires_grad_(True)
    w1 = _q(h((Hd, C), nm + "w1", math.sqrt(3.0 / C)), dtype).requires_grad_(True)
    b1 = h((Hd,), nm + "b1", 0.3).requires_grad_(True)
    w2 = _q(h((C, Hd), nm + "w2", math.sqrt(3.0 / Hd)), dtype).requires_grad_(True)
    b2 = h((C,), nm + "b2", 0.3).requires_grad_(True)
    gy = _q(h((M, C), nm + "g", 1.0), dtype)
    hm = F.gelu(F.linear(x, w1, b1)) * m1.float() / (1 - p)
    if dtype == torch.bfloat16:
        hm = hm + (hm.detach().bfloat16().float() - hm.detach())       # the kernel stores the hidden activation in bf16 (straight-through)
    ref = x + alpha * (F.linear(hm, w2, b2) * m2.float() / (1 - p))
    ref.backward(gy)
    dv = [t.detach().to(DEV).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    dv[0] = x.detach().to(dtype).to(DEV).requires_grad_(True)
    y = ops.ffn(dv[0], dv[1], dv[2], dv[3], dv[4], alpha, drop_p=p, seed1=s1, seed2=s2)
    y.backward(gy.to(dtype).to(DEV))
    tol = TOL[dtype]
    assert maxrel(y.float().cpu(), ref) < tol, "y"
    for what, a, b in (("dx", dv[0].grad, x.grad), ("dw1", dv[1].grad, w1.grad), ("db1", dv[2].grad, b1.grad),
                       ("dw2", dv[3].grad, w2.grad), ("db2", dv[4].grad, b2.grad)):
        assert maxrel(a.float().cpu(), b) < tol * 2, (what, maxrel(a.float().cpu(), b))
    # exactly the dropped hidden units / outputs carry no gradient: db2 only sums kept outputs (checked above through the
    # reference); and the output equals the input wherever the second mask dropped
    dropped = ~m2
    assert torch.equal(y.float().cpu()[dropped], x.detach()[dropped])


# ---------------------------------------------------------------- optimizer
def test_sumsq_adamw_clip(psg):
    from pokemon_sprite_generator_amd import FusedAdamW, GradArena
    torch.manual_seed(1)
    shapes = [(320, 64, 3, 3), (1280,), (77, 13), (5,)]
    ps_ref = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    ps = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ps_ref]
    arena = GradArena(ps)
    opt_ref = torch.optim.AdamW(ps_ref, lr=3e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01)
    opt = FusedAdamW(ps, lr=3e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01)
    for step in range(4):
        gs = [torch.randn(s) * (3.0 if step % 2 == 0 else 0.01) for s in shapes]
        arena.zero()
        for p, pr, g in zip(ps, ps_ref, gs):
            pr.grad = g.clone()
            p.grad.copy_(g.to(DEV))
        total = torch.nn.utils.clip_grad_norm_(ps_ref, max_norm=1.0)
        nsq = arena.grad_norm_sq()
        assert abs(float(nsq.sqrt().item()) - float(total)) / float(total) < 1e-5
        opt_ref.step()
        opt.step(normsq=nsq, max_norm=1.0)
        for p, pr in zip(ps, ps_ref):
            assert maxrel(p.detach().cpu(), pr.detach()) < 2e-6, f"step {step}"
    sd = opt.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}      # torch.optim.AdamW wire format
    arena.release()


def test_flat_arena_adamw_matches_torch(psg):
    """ParamArena (flat masters, 3x3 weights OHWI) + GradArena + single-launch FusedAdamW == torch.optim.AdamW
    after clip_grad_norm_; optimizer state_dict round-trips in torch's wire format (checkpoint resume)."""
    from pokemon_sprite_generator_amd import FusedAdamW, GradArena, ParamArena, ops
    torch.manual_seed(2)
    shapes = [(64, 32, 3, 3), (130,), (96, 64, 1, 1), (77, 13), (5,)]
    ps_ref = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    ps = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ps_ref]
    pa = ParamArena(ps)
    assert ops.weight_layout(ps[0]) == ops.W_OHWI and ps[2].is_contiguous() and ps[0].shape == (64, 32, 3, 3)
    for p, pr in zip(ps, ps_ref):
        assert torch.equal(p.detach().cpu(), pr.detach())
    ga = GradArena(ps)
    assert ps[0].grad.stride() == ps[0].stride()
    kw = dict(lr=3e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01)
    opt_ref = torch.optim.AdamW(ps_ref, **kw)
    opt = FusedAdamW(ps, param_arena=pa, grad_arena=ga, **kw)

    def run(opt, steps, s0):
        for step in range(s0, s0 + steps):
            g0 = torch.Generator().manual_seed(100 + step)
            gs = [torch.randn(s, generator=g0) * (3.0 if step % 2 == 0 else 0.01) for s in shapes]
            ga.zero()
            for p, pr, g in zip(ps, ps_ref, gs):
                pr.grad = g.clone()
                p.grad.copy_(g.to(DEV))
            torch.nn.utils.clip_grad_norm_(ps_ref, max_norm=1.0)
            opt_ref.step()
            opt.step(normsq=ga.grad_norm_sq(), max_norm=1.0)
            for p, pr in zip(ps, ps_ref):
                assert maxrel(p.detach().cpu(), pr.detach()) < 2e-6, f"step {step}"

    run(opt, 3, 0)
    sd = opt.state_dict()
    assert float(sd["state"][0]["step"]) == 3.0 and sd["state"][0]["exp_avg"].shape == (64, 32, 3, 3)
    opt2 = FusedAdamW(ps, param_arena=pa, grad_arena=ga, **kw)            # resume into a fresh optimizer
    opt2.load_state_dict({"state": {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd["state"].items()},
                          "param_groups": sd["param_groups"]})
    run(opt2, 2, 3)
    ga.release()


@pytest.mark.parametrize("M,K,N", [(1000, 192, 256), (777, 640, 320), (300, 128, 64), (8200, 256, 512)])
def test_conv_epilogue_kinds_bitwise(psg, M, K, N):
    """The bf16 staged epilogue exists once per kind (plain / dropout / GELU / GELU+dropout / saved-derivative product), with
    the residual joining in the row-major domain and `preact` + y staged side by side; PSG_CONV_GENERIC_EPILOGUE runs the
    same launch through the run-time, one-step form.  The two must agree BIT FOR BIT - y and, where written, `preact` -
    over ragged M (tile edges), the 128x128 / 128x160 / 64x64 tiles and every operand combination the U-Net issues."""
    import ctypes as C
    from pokemon_sprite_generator_amd import ops, _lib
    from pokemon_sprite_generator_amd._lib import ConvDesc
    lib = _lib.init(0)
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(M + K + N)
    r = lambda *s: torch.randn(*s, generator=g)
    x = r(M, K).bfloat16().to(DEV)
    w = (r(N, K) * 0.05).to(DEV)
    wf, wd = ops.WeightCache.get(w, torch.bfloat16, True)
    bias = r(N).to(DEV)
    res = r(M, N).bfloat16().to(DEV)
    sav = r(M, N).bfloat16().to(DEV)
    rowadd = r(7, N).bfloat16().to(DEV)

    def run(generic, **kw):
        y = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
        pre = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
        d = ConvDesc()
        d.dtype = _lib.PSG_BF16
        d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = M, 1, 1, K, 1, 1, N
        d.ksize, d.stride, d.pad, d.transposed, d.act = 1, 1, 0, 0, kw.get("act", 0)
        d.alpha, d.drop_p, d.drop_seed = kw.get("alpha", 1.0), kw.get("drop", 0.0), 99
        d.flags = kw.get("flags", 0) | (_lib.CONV_GENERIC_EPILOGUE if generic else 0)
        d.ldx, d.ldy, d.ldw = K, N, 0
        d.x, d.w, d.y = x.data_ptr(), wf.data_ptr(), y.data_ptr()
        if kw.get("bias"): d.bias = bias.data_ptr()
        if kw.get("res"): d.residual, d.ld_residual = res.data_ptr(), N
        if kw.get("pre"): d.preact, d.ld_preact = pre.data_ptr(), N
        if kw.get("dact"): d.dact_u, d.ld_dact = sav.data_ptr(), N
        _lib.check(lib.psg_conv_fwd(C.byref(d), st), "psg_conv_fwd")
        torch.cuda.synchronize()
        return y, pre

    variants = [dict(), dict(bias=1), dict(bias=1, res=1), dict(bias=1, res=1, drop=0.1, alpha=0.5), dict(bias=1, drop=0.1),
                dict(bias=1, act=_lib.ACT_GELU), dict(bias=1, act=_lib.ACT_GELU, res=1), dict(bias=1, act=_lib.ACT_GELU, drop=0.1, pre=1, flags=_lib.CONV_SAVE_DACT),
                dict(bias=1, act=_lib.ACT_GELU, pre=1, flags=_lib.CONV_SAVE_DACT), dict(bias=1, pre=1), dict(bias=1, pre=1, res=1),
                dict(dact=1, flags=_lib.CONV_DACT_MUL, alpha=0.7), dict(bias=1, act=_lib.ACT_SILU, res=1)]
    for kw in variants:
        ya, pa = run(False, **kw)
        yb, pb = run(True, **kw)
        assert torch.equal(ya, yb), f"y differs for {kw}"
        assert torch.equal(pa, pb), f"preact differs for {kw}"
        assert torch.isfinite(ya.float()).all()


# (B, H, Cin, Cout, ks): shapes whose (tap, ci) axis tiles by 256 columns with < 5 % waste, so the weight gradient takes the
# wide 128 x 256 tile (32-pixel K steps): exact tiling, a q tail (Cin 248: 2232 of 2304 columns), a Cout tail (192 = 1.5
# row tiles), pixel counts that are not multiples of 32, the three map widths with borders, pointwise layers, batch large
# enough for several splits, and a row stride wider than Cin (a concat slice)
WIDE_CASES = [(3, 14, 256, 128, 3), (5, 7, 248, 192, 3), (2, 27, 256, 128, 3), (37, 4, 512, 256, 3), (3, 9, 1280, 128, 1),
              (64, 14, 256, 256, 3), (1, 5, 256, 64, 3), (130, 7, 768, 128, 1),
              # a 1 x 1 layer with 30 row tiles, and Cout = 320 (stays on the 160 x 128 tiles)
              (5, 14, 1280, 3840, 1), (9, 27, 256, 320, 3)]


@pytest.mark.parametrize("layout", ["oihw", "ohwi"])
@pytest.mark.parametrize("B,H,Cin,Cout,ks", WIDE_CASES)
def test_wgrad_wide_tiles(psg, B, H, Cin, Cout, ks, layout):
    from pokemon_sprite_generator_amd import ops
    dtype = torch.bfloat16
    pad = 1 if ks == 3 else 0
    name = f"wide{B}.{H}.{Cin}.{Cout}.{ks}"
    x = _q(h((B, Cin, H, H), name + "x", 1.2), dtype)
    w = _q(h((Cout, Cin, ks, ks), name + "w", math.sqrt(3.0 / (Cin * ks * ks))), dtype).requires_grad_(True)
    b = h((Cout,), name + "b", 0.2).requires_grad_(True)
    ref = F.conv2d(x, w, b, padding=pad)
    gy = _q(h(tuple(ref.shape), name + "gy", 1.0), dtype)
    ref.backward(gy)
    # x as a slice of a wider buffer (row stride Cin + 64), like the decoder's concat halves
    wide_buf = torch.zeros((B, H, H, Cin + 64), dtype=dtype, device=DEV)
    wide_buf[..., :Cin] = to_cl(x, dtype).to(DEV)
    xd = wide_buf[..., :Cin]
    wd = w.detach().to(DEV)
    if layout == "ohwi" and ks == 3:
        wd = wd.contiguous(memory_format=torch.channels_last)
    wd = torch.nn.Parameter(wd)
    bd = torch.nn.Parameter(b.detach().to(DEV))
    y = ops.conv2d(xd, wd, bd)
    y.backward(to_cl(gy, dtype).to(DEV))
    assert maxrel(wd.grad.cpu(), w.grad) < 1e-4 and rel_l2(wd.grad.cpu(), w.grad) < 1e-5, "wgrad"
    assert maxrel(bd.grad.cpu(), b.grad) < 1e-4, "bias gradient"
    # accumulate path (a weight used twice through a gradient sink adds into the first result)
    from pokemon_sprite_generator_amd import GradArena
    arena = GradArena([wd, bd])
    arena.zero()
    y = ops.conv2d(xd, wd, bd) + ops.conv2d(xd, wd, bd)
    y.backward(to_cl(gy, dtype).to(DEV))
    arena.finalize()
    torch.cuda.synchronize()
    assert maxrel(wd.grad.cpu(), 2 * w.grad) < 1e-4, "wgrad accumulate"
    assert maxrel(bd.grad.cpu(), 2 * b.grad) < 1e-4, "bias accumulate"
    arena.release()


# (B, H, Cin, Cout, ks, stride): small output grids with a long K axis - the launches psg_conv_fwd splits along K when it
# is offered a workspace (sampling at 64 samples on the 7x7 / 4x4 levels, training batches of 2-4)
SPLITK_CASES = [(2, 7, 1280, 1280, 3, 1), (4, 4, 2560, 1280, 3, 1), (3, 7, 2560, 640, 1, 1), (1, 14, 640, 640, 3, 1), (6, 4, 1280, 2560, 1, 1),
                (2, 14, 320, 640, 3, 2)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Cin,Cout,ks,stride", SPLITK_CASES)
def test_conv_splitk_matches_unsplit(psg, dtype, B, H, Cin, Cout, ks, stride):
    """Split-K forward and data gradient (fp32 partial tiles + finishing kernel with the run-time epilogue) against the
    unsplit launch of the same operands and against the CPU convolution: bias, per-sample add, residual, SiLU, gate."""
    from pokemon_sprite_generator_amd import ops
    pad = 1 if ks == 3 else 0
    name = f"sk{B}.{H}.{Cin}.{Cout}.{ks}.{stride}"
    x = _q(h((B, Cin, H, H), name + "x", 1.2), dtype).requires_grad_(True)
    w = _q(h((Cout, Cin, ks, ks), name + "w", math.sqrt(3.0 / (Cin * ks * ks))), dtype)
    b = h((Cout,), name + "b", 0.2)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    Ho = ref.shape[-1]
    ra = _q(h((B, Cout), name + "ra", 0.5), dtype)
    res = _q(h((B, Cout, Ho, Ho), name + "res", 1.0), dtype)
    ref = res + 0.7 * F.silu(ref + ra[:, :, None, None])
    gy = _q(h(tuple(ref.shape), name + "gy", 1.0), dtype)
    ref.backward(gy)
    outs = {}
    saved = ops._SPLITK
    try:
        for split in (False, True):
            ops._SPLITK = split
            n0 = ops.SplitKStats.launches
            xd = to_cl(x.detach(), dtype).to(DEV).requires_grad_(True)
            y = ops.conv2d(xd, w.to(DEV), b.to(DEV), stride=stride, rowadd=ra.to(dtype).to(DEV), residual=to_cl(res, dtype).to(DEV),
                           act=ops.ACT_SILU, alpha=0.7)
            y.backward(to_cl(gy, dtype).to(DEV))
            outs[split] = (from_cl(y.detach().cpu()), from_cl(xd.grad.cpu()), ops.SplitKStats.launches - n0)
    finally:
        ops._SPLITK = saved
    assert outs[False][2] == 0
    assert outs[True][2] >= 1, "the launch was not offered / did not want a split-K workspace"
    tol = TOL[dtype]
    for i, what in ((0, "forward"), (1, "dgrad")):
        refv = ref.detach() if i == 0 else x.grad
        assert maxrel(outs[True][i], refv) < tol, what + " (split) vs CPU"
        # same products, another fp32 summation order (and one rounding to the output dtype)
        assert maxrel(outs[True][i], outs[False][i]) < (2e-5 if dtype == torch.float32 else 1.6e-2), what + " split vs unsplit"


# (B, H, Cin, Cout, ks): Cout % 320 == 0, the (tap, ci) axis a multiple of 192, >= 40 tiles of 320 x 192 and NO bias gradient in
# the launch: the software-pipelined one-wave-per-SIMD kernel (ring of four stages).  Pixel counts chosen to give 1, 2, 3, 4, 5,
# 7 and many K steps (prologue / steady state / tail paths), M not a multiple of 32, borders at every map width, 1x1 layers.
PIPE_CASES = [(2, 4, 1280, 640, 3), (4, 4, 1280, 640, 3), (6, 4, 1024, 320, 3), (2, 7, 1280, 640, 3), (3, 7, 1024, 320, 3), (1, 14, 1280, 640, 3),
              (5, 14, 1024, 320, 3), (2, 27, 1024, 320, 3), (3, 9, 1920, 1280, 1), (33, 7, 1280, 640, 3),
              (3, 9, 1888, 1280, 1)]          # (the last: a ragged last q tile, 32 of its 192 columns beyond Cin)


@pytest.mark.parametrize("with_bias", [False, True], ids=["nobias", "bias"])
@pytest.mark.parametrize("B,H,Cin,Cout,ks", PIPE_CASES)
def test_wgrad_pipe_tiles(psg, B, H, Cin, Cout, ks, with_bias):
    from pokemon_sprite_generator_amd import ops
    dtype = torch.bfloat16
    pad = 1 if ks == 3 else 0
    name = f"pipe{B}.{H}.{Cin}.{Cout}.{ks}"
    x = _q(h((B, Cin, H, H), name + "x", 1.2), dtype)
    w = _q(h((Cout, Cin, ks, ks), name + "w", math.sqrt(3.0 / (Cin * ks * ks))), dtype).requires_grad_(True)
    b = h((Cout,), name + "b", 0.2).requires_grad_(True)
    ref = F.conv2d(x, w, b if with_bias else None, padding=pad)
    gy = _q(h(tuple(ref.shape), name + "gy", 1.0), dtype)
    ref.backward(gy)
    xd = to_cl(x, dtype).to(DEV)
    for layout in ("oihw", "ohwi"):
        wd = w.detach().to(DEV)
        if layout == "ohwi" and ks == 3:
            wd = wd.contiguous(memory_format=torch.channels_last)
        wd = torch.nn.Parameter(wd)
        bd = torch.nn.Parameter(b.detach().to(DEV)) if with_bias else None
        y = ops.conv2d(xd, wd, bd)
        y.backward(to_cl(gy, dtype).to(DEV))
        assert maxrel(wd.grad.cpu(), w.grad) < 1e-4 and rel_l2(wd.grad.cpu(), w.grad) < 1e-5, f"wgrad ({layout})"
        if with_bias:
            assert maxrel(bd.grad.cpu(), b.grad) < 1e-4, f"bias gradient ({layout})"


def test_available_cus_changes_the_plan_not_the_result(psg):
    """psg_set_available_cus (the planning input of the data-parallel overlap): with fewer CUs the
    choosers pick other tiles / split counts - the results stay those of the same GEMM (fp32 accumulation order aside) - and
    a CU-masked stream (psg_stream_create_cu_mask) runs the same launches."""
    from pokemon_sprite_generator_amd import _lib, ops
    lib = _lib.init(0)
    torch.manual_seed(3)
    x = torch.randn(64, 14, 14, 640, device=DEV).bfloat16().requires_grad_(True)
    w = (torch.randn(640, 640, 3, 3, device=DEV) * 0.02).requires_grad_(True)
    gy = torch.randn(64, 14, 14, 640, device=DEV).bfloat16()

    def run():
        xs, ws = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
        y = ops.conv2d(xs, ws)
        y.backward(gy)
        torch.cuda.synchronize()
        return y.detach().float(), xs.grad.float(), ws.grad.float()

    ref = run()
    sp = C.c_void_p()
    try:
        _lib.set_available_cus(0, 96)
        got = run()
        _lib.check(lib.psg_stream_create_cu_mask(96, C.byref(sp)), "psg_stream_create_cu_mask")
        ext = torch.cuda.ExternalStream(sp.value, device=torch.device(DEV, 0))
        ext.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ext):
            masked = run()
        torch.cuda.current_stream().wait_stream(ext)
    finally:
        _lib.set_available_cus(0, 0)
        torch.cuda.synchronize()
        if sp.value:
            _lib.check(lib.psg_stream_destroy(sp), "psg_stream_destroy")
    for a, b, c in zip(ref, got, masked):
        assert rel_l2(b, a) < 2e-3 and rel_l2(c, a) < 2e-3, (rel_l2(b, a), rel_l2(c, a))
    assert lib.psg_set_available_cus(5) != 0                      # (out of range: an error code, not a crash)

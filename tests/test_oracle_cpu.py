"""CPU suite: the oracle against the committed golden fixtures (generated from the reference source),
the plain-C oracle against the torch oracle, and host-side logic.  No GPU, no reference needed."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import cases, hashgen, unet_oracle as O
from tests.util import check_digest, h, maxrel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["betas", "alphas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"]


def test_hashgen_is_stable():
    """The generator is pure integer arithmetic: pin a few values forever."""
    u = hashgen.uniform((5,), 1234, 7)
    assert u.dtype == torch.float32 and float(u.abs().max()) < 1.0
    v = hashgen.uniform((1000,), 1234, 7)
    assert torch.equal(u, v[:5])
    assert abs(float(v.mean())) < 0.1 and 0.5 < float(v.std()) < 0.65
    assert hashgen.name_id("input.x") == 0x1C7E1F0B or isinstance(hashgen.name_id("input.x"), int)
    big = hashgen.uniform((3, 1 << 12), 9, 9, chunk=1000)          # chunking does not change values
    assert torch.equal(big, hashgen.uniform((3, 1 << 12), 9, 9))


def test_schedule_tables_bit_exact(golden):
    g = golden("schedule.npz")
    tc, tl, t250 = O.cosine_clipped_tables(), O.linear_tables(), O.cosine_clipped_tables(250, 0.0005, 0.03)
    for n in NAMES:
        assert np.array_equal(tc[n].numpy(), g["cos_" + n]), n
        assert np.array_equal(tl[n].numpy(), g["lin_" + n]), n
        assert np.array_equal(t250[n].numpy(), g["cos250_" + n]), n
    # known answers recorded in SURVEY.md §8 a-1
    assert tc["sqrt_alphas_cumprod"][500].view(torch.int32).item() == 0x3F339536
    assert int((tc["betas"] == tc["betas"].min()).sum()) == 13 and int((tc["betas"] == tc["betas"].max()).sum()) == 98


def test_product_scheduler_tables_bit_exact(golden):
    """Host logic of the product: NoiseScheduler builds the same five tables (no GPU involved)."""
    from pokemon_sprite_generator_amd import NoiseScheduler
    g = golden("schedule.npz")
    s = NoiseScheduler()
    for n in NAMES:
        assert np.array_equal(getattr(s, n).numpy(), g["cos_" + n]), n
    s2 = NoiseScheduler(250, 0.0005, 0.03)
    for n in NAMES:
        assert np.array_equal(getattr(s2, n).numpy(), g["cos250_" + n]), n
    assert s.num_timesteps == 1000
    c1, c2, sg = s.step_tables("cpu")
    a, b, c = O.ddpm_step_coeffs(O.cosine_clipped_tables(), 500)
    assert c1[500] == a and c2[500] == b and sg[500] == c


def test_add_noise_oracle_golden(golden):
    g = golden("add_noise.npz")
    x0 = torch.clamp(hashgen.uniform((6, 8, 9, 9), cases.INPUT_SEED, hashgen.name_id("an.x0")) * 3.5, -3.0, 3.0)
    nz = hashgen.uniform((6, 8, 9, 9), cases.INPUT_SEED, hashgen.name_id("an.noise")) * 2.0
    t = torch.from_numpy(g["t"])
    assert np.array_equal(O.add_noise(x0, nz, t, O.cosine_clipped_tables()).numpy(), g["out"])
    assert np.array_equal(O.add_noise(x0, nz, t, O.linear_tables()).numpy(), g["out_lin"])
    bad = nz.clone()
    bad[2, 3, 4, 5] = float("inf")
    assert np.array_equal(O.add_noise(x0, bad, t, O.cosine_clipped_tables()).numpy(), g["out_fallback"])


def test_c_oracle_matches_torch_oracle():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libpsg_oracle.so"))
    lib.oracle_smooth_l1_f32.restype = C.c_double
    fp = lambda a: a.numpy().ctypes.data_as(C.c_void_p)
    x0, nz = h((4, 8, 27, 27), "c.x0", 3.5), h((4, 8, 27, 27), "c.nz", 1.0)
    t = torch.tensor([0, 500, 999, 37], dtype=torch.int64)
    tb = O.cosine_clipped_tables()
    out, fb = torch.empty_like(x0), torch.empty_like(x0)
    bad = lib.oracle_noise_add_f32(fp(x0), fp(nz), t.numpy().ctypes.data_as(C.c_void_p), fp(tb["sqrt_alphas_cumprod"]),
                                   fp(tb["sqrt_one_minus_alphas_cumprod"]), fp(out), fp(fb), C.c_int64(4), C.c_int64(5832), 1)
    xc = torch.clamp(x0, -3.0, 3.0)
    assert bad == 0 and torch.equal(out, O.add_noise(xc, nz, t, tb)) and torch.equal(fb, xc + 0.1 * nz)
    # ddpm update
    x, eps, z = h((2, 8, 27, 27), "c.x", 2.0), h((2, 8, 27, 27), "c.e", 1.0), h((2, 8, 27, 27), "c.z", 1.0)
    c1, c2, sg = O.ddpm_step_coeffs(tb, 450)
    ref = c1 * (x - c2 * eps) + sg * z
    xx = x.clone()
    lib.oracle_ddpm_update_f32(fp(xx), fp(eps), fp(z), C.c_float(float(c1)), C.c_float(float(c2)), C.c_float(float(sg)), 1, C.c_int64(x.numel()))
    assert torch.equal(xx, ref)
    # SmoothL1
    p, q = h((3, 100), "c.p", 1.0), h((3, 100), "c.q", 1.0)
    pr = p.clone().requires_grad_(True)
    l = O.smooth_l1(pr, q, 0.1)
    l.backward()
    g = torch.empty_like(p)
    lc = lib.oracle_smooth_l1_f32(fp(p), fp(q), fp(g), C.c_float(0.1), C.c_int64(p.numel()))
    assert abs(lc - float(l)) < 1e-6 and maxrel(g, pr.grad) < 1e-6
    assert abs(float(torch.nn.SmoothL1Loss(beta=0.1)(p, q)) - float(l)) < 1e-7


def test_block_oracle_golden(golden):
    """Oracle restatement of TimestepEmbedding / ResBlock / CrossAttentionBlock vs reference-generated fixtures."""
    g = golden("blocks.npz")
    shapes = {"time_embed.emb_coeff": (64,), "time_embed.time_mlp.0.weight": (512, 128), "time_embed.time_mlp.0.bias": (512,),
              "time_embed.time_mlp.2.weight": (512, 512), "time_embed.time_mlp.2.bias": (512,),
              "time_embed.time_mlp.4.weight": (128, 512), "time_embed.time_mlp.4.bias": (128,)}
    sd = hashgen.fill_unet_state(shapes, cases.WEIGHT_SEED, "stress")
    out = O.timestep_embedding(torch.tensor(cases.TIME_EMBED_T), sd)
    assert maxrel(out, torch.from_numpy(g["time_embed_out"])) < 1e-5
    for name, cin, cout, hw, b in cases.RESBLOCK_CASES:
        shapes = {"rb.norm1.weight": (cin,), "rb.norm1.bias": (cin,), "rb.conv1.weight": (cout, cin, 3, 3), "rb.conv1.bias": (cout,),
                  "rb.time_proj.weight": (cout, 128), "rb.time_proj.bias": (cout,), "rb.text_proj.weight": (cout, 256),
                  "rb.text_proj.bias": (cout,), "rb.norm2.weight": (cout,), "rb.norm2.bias": (cout,),
                  "rb.conv2.weight": (cout, cout, 3, 3), "rb.conv2.bias": (cout,)}
        if cin != cout:
            shapes.update({"rb.skip_conv.weight": (cout, cin, 1, 1), "rb.skip_conv.bias": (cout,)})
        sd = hashgen.fill_unet_state(shapes, cases.WEIGHT_SEED, "stress")
        x = hashgen.uniform((b, cin, hw, hw), cases.INPUT_SEED, hashgen.name_id(name + ".x")) * 1.7
        temb = hashgen.uniform((b, 128), cases.INPUT_SEED, hashgen.name_id(name + ".temb"))
        pooled = hashgen.uniform((b, 256), cases.INPUT_SEED, hashgen.name_id(name + ".pooled"))
        y = O.resblock(x, temb, pooled, sd, "rb.")
        assert maxrel(y, torch.from_numpy(g[name + "_y"])) < 1e-5, name


def test_attention_oracle_golden(golden):
    g = golden("blocks.npz")
    name, ch, heads, hw, b, seq = cases.ATTN_CASES[0]
    E = ch
    shapes = {"ab.norm1.weight": (E,), "ab.norm1.bias": (E,), "ab.norm2.weight": (E,), "ab.norm2.bias": (E,),
              "ab.self_attn.in_proj_weight": (3 * E, E), "ab.self_attn.in_proj_bias": (3 * E,),
              "ab.self_attn.out_proj.weight": (E, E), "ab.self_attn.out_proj.bias": (E,),
              "ab.cross_attn.in_proj_weight": (3 * E, E), "ab.cross_attn.in_proj_bias": (3 * E,),
              "ab.cross_attn.out_proj.weight": (E, E), "ab.cross_attn.out_proj.bias": (E,),
              "ab.text_proj.weight": (E, 256), "ab.text_proj.bias": (E,),
              "ab.ffn.0.weight": (2 * E, E), "ab.ffn.0.bias": (2 * E,), "ab.ffn.3.weight": (E, 2 * E), "ab.ffn.3.bias": (E,)}
    sd = hashgen.fill_unet_state(shapes, cases.WEIGHT_SEED, "stress")
    x = hashgen.uniform((b, ch, hw, hw), cases.INPUT_SEED, hashgen.name_id(name + ".x")) * 1.7
    text = hashgen.uniform((b, seq, 256), cases.INPUT_SEED, hashgen.name_id(name + ".text")) * 1.7
    y = O.cross_attention_block(x, text, sd, "ab.", heads)
    check_digest(y, g[name + "_y_d"], g[name + "_y_s"], 1e-5, name)


def test_smooth_l1_and_adamw_restatements():
    p, q = h((7, 33), "s.p", 1.0), h((7, 33), "s.q", 1.0)
    assert abs(float(O.smooth_l1(p, q, 0.1)) - float(torch.nn.SmoothL1Loss(beta=0.1)(p, q))) < 1e-7
    w = torch.nn.Parameter(h((5, 9), "a.w", 1.0))
    opt = torch.optim.AdamW([w], lr=3e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01)
    pp, m, v = w.detach().clone(), torch.zeros(5, 9), torch.zeros(5, 9)
    for step in range(1, 4):
        gr = h((5, 9), f"a.g{step}", 1.0)
        w.grad = gr.clone()
        opt.step()
        pp, m, v = O.adamw_update(pp, gr, m, v, step, 3e-4, 0.9, 0.999, 1e-6, 0.01)
        assert maxrel(pp, w.detach()) < 1e-6
    assert O.clip_coef(4.0, 1.0) == pytest.approx(1.0 / (4.0 + 1e-6)) and O.clip_coef(0.5, 1.0) == 1.0
    assert O.ddpm_timesteps(1000, True)[:3] == [950, 900, 850] and len(O.ddpm_timesteps(1000, False)) == 1000


@pytest.mark.timeout(900)
def test_full_unet_oracle_golden(golden):
    """Full-width (640 M parameter) oracle forward vs the reference fixture, B=1, stress weights."""
    g = golden("unet_full.npz")
    name, mode, b, ts, heads = cases.UNET_CASES[1]
    shapes = _full_shapes()
    sd = hashgen.fill_unet_state(shapes, cases.WEIGHT_SEED, mode)
    x, t, text = hashgen.unet_inputs(b, cases.INPUT_SEED, t=ts)
    with torch.no_grad():
        y = O.unet_forward(sd, x, t, text, heads)
    assert maxrel(y, torch.from_numpy(g[name + "_eps"])) < 1e-5


def _full_shapes():
    """Key/shape map of the 479-entry reference state_dict, from the product's own container modules."""
    import pokemon_sprite_generator_amd as psg
    with torch.device("meta"):
        u = psg.UNet()
    return {k: tuple(v.shape) for k, v in u.state_dict().items()}


def test_product_state_dict_layout():
    shapes = _full_shapes()
    assert len(shapes) == 479
    n = sum(int(np.prod(s)) for k, s in shapes.items() if not k.endswith("emb_coeff"))
    assert n == 640488456
    assert shapes["enc_block1.0.attn_block.cross_attn.in_proj_weight"] == (1920, 640)
    assert shapes["dec_block0.1.res_block.skip_conv.weight"] == (320, 640, 1, 1)
    assert shapes["middle_block.attn_block.ffn.3.weight"] == (1280, 2560)
    assert "enc_block0.0.attn_block.norm1.weight" not in shapes and "dec_block0.0.attn_block.norm1.weight" not in shapes


def test_inference_oracle_golden(golden):
    """f-3: the oracle's restatement of stage 3's NoiseScheduler (final_trainer.py:19-81) reproduces the outputs of the
    reference's own class (fixture written by oracle/make_golden_inference.py) bit for bit, and so do the host-side
    tables of the product's LinearNoiseScheduler (the per-element update itself needs the GPU: tests/test_unet_gpu.py)."""
    import math
    import pokemon_sprite_generator_amd as psg
    from oracle import cases, hashgen
    g = golden("inference.npz")
    tb = O.final_linear_tables()
    sch = psg.LinearNoiseScheduler()
    names = ("betas", "alphas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas",
             "posterior_variance")
    for n in names:      # (torch's vectorised CPU linspace / sqrt differ in the last bit between host CPUs: 1 ulp)
        assert np.allclose(tb[n].numpy(), g["final_" + n], rtol=4e-7, atol=0), n
        assert torch.equal(getattr(sch, n), tb[n]), n            # product tables == oracle tables on this host, bit for bit
    tb = {n: torch.from_numpy(g["final_" + n]) for n in names}  # bit-exact checks below run on the fixture's tables
    x0 = hashgen.uniform((5, 8, 9, 9), cases.INPUT_SEED, hashgen.name_id("fin.x0")) * 2.5
    nz = hashgen.uniform((5, 8, 9, 9), cases.INPUT_SEED, hashgen.name_id("fin.noise")) * 2.0
    t = torch.from_numpy(g["final_add_noise_t"])
    assert np.array_equal(O.final_add_noise(x0, nz, t, tb).numpy(), g["final_add_noise"])
    for ts in (0, 1, 500, 999):
        z = hashgen.uniform(tuple(x0.shape), cases.INPUT_SEED, hashgen.name_id(f"spt{ts}.z0")) * math.sqrt(3.0)
        assert np.array_equal(O.sample_previous_timestep(x0, nz, ts, tb, z).numpy(), g[f"final_prev_t{ts}"]), ts
    gt, here = O.gradio_tables(), O.final_linear_tables()
    assert torch.equal(gt["betas"], here["betas"]) and torch.equal(gt["alphas_cumprod"], here["alphas_cumprod"])


def test_vae_oracle_golden(golden):
    """f-4: the VAE oracle (oracle/vae_oracle.py) reproduces the reference module's outputs (fixture written by
    oracle/make_golden_vae.py from src/models/vae_decoder.py itself): encoder tensors and the small attention-block pins."""
    from oracle import vae_oracle as V
    from oracle.make_golden_vae import SEED_IN, SEED_W, vae_inputs
    import pokemon_sprite_generator_amd as psg
    g = golden("vae.npz")
    enc = psg.VAEEncoder(3, 8)                                   # parameter container: names / shapes of the reference
    esd = hashgen.fill_unet_state({k: tuple(v.shape) for k, v in enc.state_dict().items()}, SEED_W, "stress")
    img, eps, _, _ = vae_inputs()
    torch.set_num_threads(8)
    lat, mu, lv = V.vae_encode(esd, img, eps)
    for got, name in ((mu, "enc_mu"), (lv, "enc_logvar"), (lat, "enc_latent")):
        assert maxrel(got, torch.from_numpy(g[name])) < 1e-5, name
    for c in (512, 32):
        blk = psg.vae.CrossAttentionBlock(c, 256)
        bsd = hashgen.fill_unet_state({"ab." + k: tuple(v.shape) for k, v in blk.state_dict().items()}, SEED_W + 2, "stress")
        x = hashgen.uniform((2, c, 6, 5), SEED_IN, hashgen.name_id(f"vae.ab{c}.x")) * 1.5
        tx = hashgen.uniform((2, 20, 256), SEED_IN, hashgen.name_id(f"vae.ab{c}.t")) * 1.5
        assert maxrel(V.cross_attention_block(x, tx, bsd, "ab."), torch.from_numpy(g[f"attn{c}_y"])) < 1e-5

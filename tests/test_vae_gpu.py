"""-m gpu: the frozen VAE either side of the U-Net path (reference src/models/vae_decoder.py) on the
MI355X kernels, against outputs of the REFERENCE module itself (tests/golden/vae.npz, oracle/make_golden_vae.py) and
against the CPU oracle on the same seeded inputs.

Tolerances: fp32 path <= 1e-3 max-rel (the north-star bar); bf16 path: rel-L2 < 3e-2 (8 significant bits, stated)."""
import math

import numpy as np
import pytest
import torch

from oracle import hashgen, vae_oracle as V
from oracle.make_golden_vae import SEED_IN, SEED_W, vae_inputs
from tests.util import check_digest, maxrel, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP32_TOL = 1e-3


@pytest.fixture(scope="module")
def psg():
    import pokemon_sprite_generator_amd as m
    from pokemon_sprite_generator_amd import _lib
    _lib.init(0)
    return m


def _fill(module, seed):
    sd = hashgen.fill_unet_state({k: tuple(v.shape) for k, v in module.state_dict().items()}, seed, "stress")
    module.load_state_dict(sd)
    return sd


def test_vae_state_dict_matches_reference_layout(psg):
    """Checkpoint interchange: a stage-1 'vae_state_dict' splits into encoder./decoder. halves with these keys and shapes."""
    vae = psg.PokemonVAE()
    sd = vae.state_dict()
    assert tuple(sd["encoder.encoder.0.weight"].shape) == (32, 3, 4, 4)
    assert tuple(sd["encoder.encoder.9.shortcut.weight"].shape) == (256, 128, 1, 1)
    assert tuple(sd["encoder.logvar_proj.weight"].shape) == (8, 512, 3, 3)
    assert tuple(sd["decoder.block2_attn.k.weight"].shape) == (256, 256)
    assert tuple(sd["decoder.block5_attn.q.weight"].shape) == (32, 32, 1, 1)
    assert tuple(sd["decoder.final_conv.2.weight"].shape) == (3, 32, 3, 3)
    # (measured on the reference's own PokemonVAE in the authoring container: 214 entries in this key order, 25,914,675 parameters)
    assert len(sd) == 214 and sum(p.numel() for p in vae.parameters()) == 25_914_675
    assert len([k for k in sd if k.startswith("encoder.")]) == 70 and len([k for k in sd if k.startswith("decoder.")]) == 144


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_vae_encoder_golden(psg, golden, dtype):
    g = golden("vae.npz")
    enc = psg.VAEEncoder(3, 8, compute_dtype=dtype)
    _fill(enc, SEED_W)
    enc = enc.to(DEV)
    img, eps, _, _ = vae_inputs()
    latent, mu, logvar = enc(img.to(DEV), eps=eps.to(DEV))
    assert latent.shape == (1, 8, 27, 27) and latent.dtype == torch.float32
    if dtype == torch.float32:
        for got, name in ((mu, "enc_mu"), (logvar, "enc_logvar"), (latent, "enc_latent")):
            assert maxrel(got, torch.from_numpy(g[name])) < FP32_TOL, name
    else:
        for got, name in ((mu, "enc_mu"), (logvar, "enc_logvar"), (latent, "enc_latent")):
            assert rel_l2(got, torch.from_numpy(g[name])) < 3e-2, name
    # the sample really is mu + eps * exp(0.5 logvar) of ITS OWN mu / logvar (vae_decoder.py:120-123; device expf vs the
    # host's exp: last-bit differences only)
    assert maxrel(latent.cpu(), mu.cpu() + eps * torch.exp(0.5 * logvar.cpu())) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_vae_decoder_golden(psg, golden, dtype):
    g = golden("vae.npz")
    dec = psg.VAEDecoder(8, 256, 3, compute_dtype=dtype)
    _fill(dec, SEED_W + 1)
    dec = dec.to(DEV)
    _, _, lat, text = vae_inputs()
    for name, tx in (("dec", text), ("dec_s20", vae_inputs(1, 20)[3])):       # S = 32 and a ragged S = 20
        image = dec(lat.to(DEV), tx.to(DEV))
        assert image.shape == (1, 3, 215, 215) and image.dtype == torch.float32
        assert float(image.abs().max()) <= 1.0                                   # tanh range
        if dtype == torch.float32:
            check_digest(image, g[name + "_d"], g[name + "_s"], FP32_TOL, name)
        else:
            f = image.detach().reshape(-1).double().cpu()[::int(g[name + "_d"][2])].float().numpy()
            ref = g[name + "_s"]
            assert np.linalg.norm(f - ref) / np.linalg.norm(ref) < 3e-2, name


@pytest.mark.parametrize("c", [512, 256, 128, 64, 32])
def test_vae_cross_attention_block_golden(psg, golden, c):
    """Every head_dim of the decoder (64, 32, 16 on the matrix cores in bf16; 8 and 4 on the VALU kernels), with the
    reference's reshape-not-transpose key / value layout (vae_decoder.py:56-57), 2 samples, 6x5 map, 20 text tokens."""
    g = golden("vae.npz")
    blk = psg.vae.CrossAttentionBlock(c, 256)
    bsd = hashgen.fill_unet_state({"ab." + k: tuple(v.shape) for k, v in blk.state_dict().items()}, SEED_W + 2, "stress")
    blk.load_state_dict({k[3:]: v for k, v in bsd.items()})
    blk = blk.to(DEV)
    x = hashgen.uniform((2, c, 6, 5), SEED_IN, hashgen.name_id(f"vae.ab{c}.x")) * 1.5
    tx = hashgen.uniform((2, 20, 256), SEED_IN, hashgen.name_id(f"vae.ab{c}.t")) * 1.5
    ref = torch.from_numpy(g[f"attn{c}_y"])
    y = blk(x.to(DEV), tx.to(DEV))
    assert maxrel(y, ref) < FP32_TOL
    blk.compute_dtype = torch.bfloat16
    assert rel_l2(blk(x.to(DEV), tx.to(DEV)), ref) < 3e-2


def test_vae_roundtrip_vs_oracle_batch2(psg):
    """encode -> decode at batch 2 on fresh inputs (not in any fixture) against the CPU oracle; PokemonVAE modes."""
    vae = psg.PokemonVAE(8, 256)
    esd, dsd = _fill(vae.encoder, 11), _fill(vae.decoder, 12)
    vae = vae.to(DEV)
    img = hashgen.uniform((2, 3, 215, 215), 99, 1)
    eps = hashgen.uniform((2, 8, 27, 27), 99, 2) * math.sqrt(3.0)
    text = hashgen.uniform((2, 32, 256), 99, 3) * math.sqrt(3.0)
    lat, mu, lv = vae.encoder(img.to(DEV), eps=eps.to(DEV))
    o_lat, o_mu, o_lv = V.vae_encode(esd, img, eps)
    assert maxrel(mu, o_mu) < FP32_TOL and maxrel(lv, o_lv) < FP32_TOL and maxrel(lat, o_lat) < FP32_TOL
    rec = vae.decode(lat, text.to(DEV))
    assert maxrel(rec, V.vae_decode(dsd, o_lat, text)) < 2 * FP32_TOL
    out = vae(img.to(DEV), text.to(DEV), mode="generate")
    assert torch.equal(out["latent"], out["mu"]) and out["reconstructed"].shape == (2, 3, 215, 215)
    out = vae(None, text.to(DEV), mode="sample")
    assert out["mu"] is None and out["reconstructed"].shape == (2, 3, 215, 215)

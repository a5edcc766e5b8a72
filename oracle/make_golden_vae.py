#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — generate tests/golden/vae.npz FROM THE REFERENCE.

src/models/vae_decoder.py imports only torch, so the reference module is loaded by file path (as oracle/make_golden.py
loads unet.py), given weights / inputs from oracle/hashgen.py, and run on the CPU; oracle/vae_oracle.py is checked against
it and OUTPUTS ONLY are written (the 215x215 image as a strided digest).

    python oracle/make_golden_vae.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import importlib.util
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, hashgen, vae_oracle as V  # noqa: E402
from oracle.make_golden import digest, maxrel  # noqa: E402

SEED_W, SEED_IN = 606, 707


def vae_inputs(b=1, seq=32):
    img = hashgen.uniform((b, 3, 215, 215), SEED_IN, hashgen.name_id("vae.img"))                       # images in [-1, 1]
    eps = hashgen.uniform((b, 8, 27, 27), SEED_IN, hashgen.name_id("vae.eps")) * math.sqrt(3.0)
    lat = hashgen.uniform((b, 8, 27, 27), SEED_IN, hashgen.name_id("vae.lat")) * math.sqrt(3.0)
    text = hashgen.uniform((b, seq, 256), SEED_IN, hashgen.name_id("vae.text")) * math.sqrt(3.0)
    return img, eps, lat, text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    torch.set_num_threads(8)
    spec = importlib.util.spec_from_file_location("ref_vae", os.path.join(args.ref, "src/models/vae_decoder.py"))
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    out, report = {}, []
    img, eps, lat, text = vae_inputs()

    enc = R.VAEEncoder(3, 8).eval()
    esd = hashgen.fill_unet_state({k: tuple(v.shape) for k, v in enc.state_dict().items()}, SEED_W, "stress")
    enc.load_state_dict(esd)
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t: eps                                   # the reference's reparameterisation draw (:121)
    try:
        with torch.no_grad():
            latent, mu, logvar = enc(img)
    finally:
        torch.randn_like = real_randn_like
    o_lat, o_mu, o_lv = V.vae_encode(esd, img, eps)
    e = max(maxrel(o_lat, latent), maxrel(o_mu, mu), maxrel(o_lv, logvar))
    report.append(f"VAEEncoder [1,3,215,215]: oracle vs reference max rel {e:.2e}; |mu| mean {float(mu.abs().mean()):.3f}, |logvar| mean {float(logvar.abs().mean()):.3f}")
    assert e < 1e-5
    out["enc_latent"], out["enc_mu"], out["enc_logvar"] = latent.numpy(), mu.numpy(), logvar.numpy()

    dec = R.VAEDecoder(8, 256, 3).eval()
    dsd = hashgen.fill_unet_state({k: tuple(v.shape) for k, v in dec.state_dict().items()}, SEED_W + 1, "stress")
    dec.load_state_dict(dsd)
    for name, tx in (("dec", text), ("dec_s20", vae_inputs(1, 20)[3])):
        with torch.no_grad():
            image = dec(lat, tx)
        o_img = V.vae_decode(dsd, lat, tx)
        e = maxrel(o_img, image)
        report.append(f"VAEDecoder ({name}, S={tx.shape[1]}) -> [1,3,215,215]: oracle vs reference max rel {e:.2e}; |image| mean {float(image.abs().mean()):.3f}")
        assert e < 1e-5
        out[name + "_d"], out[name + "_s"] = digest(image, 8192)
    # block-level pins with small shapes (every head_dim of the decoder: 64, 32, 16, 8, 4)
    for c in (512, 256, 128, 64, 32):
        blk = R.CrossAttentionBlock(c, 256).eval()
        bsd = hashgen.fill_unet_state({"ab." + k: tuple(v.shape) for k, v in blk.state_dict().items()}, SEED_W + 2, "stress")
        blk.load_state_dict({k[3:]: v for k, v in bsd.items()})
        x = hashgen.uniform((2, c, 6, 5), SEED_IN, hashgen.name_id(f"vae.ab{c}.x")) * 1.5
        tx = hashgen.uniform((2, 20, 256), SEED_IN, hashgen.name_id(f"vae.ab{c}.t")) * 1.5
        with torch.no_grad():
            y = blk(x, tx)
        assert maxrel(V.cross_attention_block(x, tx, bsd, "ab."), y) < 1e-5
        out[f"attn{c}_y"] = y.numpy()
    report.append("VAE CrossAttentionBlock C in {512,256,128,64,32}: oracle == reference within 1e-5")

    np.savez_compressed(os.path.join(args.out, "vae.npz"), **out)
    with open(os.path.join(args.out, "REPORT_vae.txt"), "w") as f:
        f.write("oracle/make_golden_vae.py — oracle restatement vs reference source, torch %s\n" % torch.__version__)
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()

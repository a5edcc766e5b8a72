#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — generate tests/golden/inference.npz FROM THE REFERENCE.

Same method as oracle/make_golden.py (kept separate so that the round-1 fixtures stay byte-identical): the reference's
own code is AST-extracted from modules whose imports are unavailable here and run on inputs / weights from
oracle/hashgen.py with `torch.randn*` replaced by the hash generator; oracle/unet_oracle.py is checked against it and
OUTPUTS ONLY are written.

  final_trainer.py:19-81    class NoiseScheduler (linear betas, sqrt_recip_alphas, posterior_variance), add_noise,
                            sample_previous_timestep
  final_trainer.py:165-212  FinalPokemonGenerator.forward(mode='generate'), run on a stand-in `self` whose text encoder
                            and VAE decoder are identities (they are out of scope) and whose unet is the REFERENCE UNet
  gradio_app.py:279-361     PokemonGradioGenerator.setup_noise_scheduler + ddpm_sample (text-only and image-conditioned)

    python oracle/make_golden_inference.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, hashgen, unet_oracle as O  # noqa: E402
from oracle.make_golden import extract, load_ref_unet, maxrel  # noqa: E402

STEPS_FINAL, STEPS_GRADIO, HEADS = 8, 7, 8


class TorchProxy:
    """torch with randn / randn_like replaced by the hash generator (named draws)."""

    def __init__(self, tag):
        self.tag, self.n = tag, 0

    def __getattr__(self, a):
        return getattr(torch, a)

    def randn(self, *shape, device=None):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        return hashgen.uniform(tuple(shape), cases.INPUT_SEED, hashgen.name_id(self.tag + ".xT")) * math.sqrt(3.0)

    def randn_like(self, x):
        z = hashgen.uniform(tuple(x.shape), cases.INPUT_SEED, hashgen.name_id(f"{self.tag}.z{self.n}")) * math.sqrt(3.0)
        self.n += 1
        return z


def noise_for(tag):
    return lambda i, shape: hashgen.uniform(tuple(shape), cases.INPUT_SEED, hashgen.name_id(f"{tag}.z{i}")) * math.sqrt(3.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    torch.set_num_threads(8)
    out, report = {}, []

    # ---------------- final_trainer.NoiseScheduler ----------------------------------------------------------------
    NSf = extract(args.ref, "src/training/final_trainer.py", "NoiseScheduler")
    ns = NSf()
    tb = O.final_linear_tables()
    for n in ("betas", "alphas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas",
              "posterior_variance"):
        assert torch.equal(getattr(ns, n), tb[n]), n
        out["final_" + n] = getattr(ns, n).numpy()
    x0 = hashgen.uniform((5, 8, 9, 9), cases.INPUT_SEED, hashgen.name_id("fin.x0")) * 2.5
    nz = hashgen.uniform((5, 8, 9, 9), cases.INPUT_SEED, hashgen.name_id("fin.noise")) * 2.0
    tt = torch.tensor([0, 1, 500, 999, 37], dtype=torch.int64)
    an = ns.add_noise(x0, nz, tt)
    assert torch.equal(an, O.final_add_noise(x0, nz, tt, tb))
    out["final_add_noise_t"], out["final_add_noise"] = tt.numpy(), an.numpy()
    for t in (0, 1, 500, 999):
        NSp = extract(args.ref, "src/training/final_trainer.py", "NoiseScheduler", extra_globals={"torch": TorchProxy(f"spt{t}")})
        got = NSp().sample_previous_timestep(x0, nz, t)
        want = O.sample_previous_timestep(x0, nz, t, tb, noise_for(f"spt{t}")(0, x0.shape))
        assert torch.equal(got, want), t
        out[f"final_prev_t{t}"] = got.numpy()
    report.append("final_trainer.NoiseScheduler: tables, add_noise, sample_previous_timestep - oracle == reference bit-exact")

    # ---------------- the reference U-Net with stress weights -------------------------------------------------------
    R = load_ref_unet(args.ref)
    torch.manual_seed(0)
    u = R.UNet(latent_dim=8, text_dim=256, time_emb_dim=128, num_heads=HEADS).eval()
    shapes = {k: tuple(v.shape) for k, v in u.state_dict().items()}
    sd = hashgen.fill_unet_state(shapes, cases.WEIGHT_SEED, "stress")
    u.load_state_dict(sd)
    for p in u.parameters():
        p.requires_grad_(False)
    _, _, text = hashgen.unet_inputs(1, cases.INPUT_SEED)
    eps_or = lambda x, t, te: O.unet_forward(sd, x, t, te, HEADS)

    # ---------------- FinalPokemonGenerator.forward (generate) -----------------------------------------------------
    fwd = extract(args.ref, "src/training/final_trainer.py", "FinalPokemonGenerator", method="forward",
                  extra_globals={"torch": TorchProxy("fgen"), "List": list})
    NSg = extract(args.ref, "src/training/final_trainer.py", "NoiseScheduler", extra_globals={"torch": TorchProxy("fgen")})
    proxy_shared = fwd.__globals__["torch"]
    sched = NSg()
    # one shared draw counter: the reference draws x_T with randn and every step's noise with randn_like
    sched.sample_previous_timestep.__func__.__globals__["torch"] = proxy_shared
    fake = types.SimpleNamespace(text_encoder=lambda texts: text, vae_decoder=lambda latent, te: latent, unet=u, noise_scheduler=sched,
                                 latent_dim=8)
    with torch.no_grad():
        x_ref = fwd(fake, ["stub"], STEPS_FINAL, "generate")
    xT = hashgen.uniform((1, 8, 27, 27), cases.INPUT_SEED, hashgen.name_id("fgen.xT")) * math.sqrt(3.0)
    trace = []
    with torch.no_grad():
        x_or = O.final_generate_latents(eps_or, tb, text, xT, noise_for("fgen"), STEPS_FINAL, trace)
    report.append(f"FinalPokemonGenerator.forward(generate, {STEPS_FINAL} steps): oracle vs reference max rel {maxrel(x_or, x_ref):.2e}")
    assert maxrel(x_or, x_ref) < 1e-4
    out["fgen_final"] = x_ref.numpy()
    for i in (0, STEPS_FINAL // 2, STEPS_FINAL - 1):
        out[f"fgen_step{i}"] = trace[i].numpy()
    # a schedule that reaches t == 0 (step_size * steps > T): the `latent - predicted_noise` branch
    proxy2 = TorchProxy("fgen0")
    fwd0 = extract(args.ref, "src/training/final_trainer.py", "FinalPokemonGenerator", method="forward",
                   extra_globals={"torch": proxy2, "List": list})
    NS0 = extract(args.ref, "src/training/final_trainer.py", "NoiseScheduler", extra_globals={"torch": proxy2})
    fake0 = types.SimpleNamespace(text_encoder=lambda texts: text, vae_decoder=lambda latent, te: latent, unet=u,
                                  noise_scheduler=NS0(num_timesteps=6), latent_dim=8)
    with torch.no_grad():
        x_ref0 = fwd0(fake0, ["stub"], 4, "generate")          # T=6, 4 steps: step_size 1 -> t = 5,4,3,2 ... use T=3 for t==0
    out["fgen_T6_final"] = x_ref0.numpy()
    xT0 = hashgen.uniform((1, 8, 27, 27), cases.INPUT_SEED, hashgen.name_id("fgen0.xT")) * math.sqrt(3.0)
    with torch.no_grad():
        x_or0 = O.final_generate_latents(eps_or, O.final_linear_tables(6), text, xT0, noise_for("fgen0"), 4)
    assert maxrel(x_or0, x_ref0) < 1e-4
    proxy3 = TorchProxy("fgen3")
    fwd3 = extract(args.ref, "src/training/final_trainer.py", "FinalPokemonGenerator", method="forward",
                   extra_globals={"torch": proxy3, "List": list})
    NS3 = extract(args.ref, "src/training/final_trainer.py", "NoiseScheduler", extra_globals={"torch": proxy3})
    fake3 = types.SimpleNamespace(text_encoder=lambda texts: text, vae_decoder=lambda latent, te: latent, unet=u,
                                  noise_scheduler=NS3(num_timesteps=3), latent_dim=8)
    with torch.no_grad():
        x_ref3 = fwd3(fake3, ["stub"], 4, "generate")          # T=3, 4 steps: t = 2, 1, 0, 0 (clamped): two t == 0 steps
    xT3 = hashgen.uniform((1, 8, 27, 27), cases.INPUT_SEED, hashgen.name_id("fgen3.xT")) * math.sqrt(3.0)
    with torch.no_grad():
        x_or3 = O.final_generate_latents(eps_or, O.final_linear_tables(3), text, xT3, noise_for("fgen3"), 4)
    assert maxrel(x_or3, x_ref3) < 1e-4
    out["fgen_T3_final"] = x_ref3.numpy()
    report.append("FinalPokemonGenerator.forward: T=6 and T=3 (t == 0 branch, clamped timestep) oracle == reference within 1e-4")

    # ---------------- gradio_app.PokemonGradioGenerator.ddpm_sample -------------------------------------------------
    for tag, init in (("grad", None), ("gradimg", hashgen.uniform((1, 8, 27, 27), cases.INPUT_SEED, hashgen.name_id("gradimg.init")) * 1.5)):
        proxy = TorchProxy(tag)
        g = {"torch": proxy, "Optional": __import__("typing").Optional}
        setup = extract(args.ref, "gradio_app.py", "PokemonGradioGenerator", method="setup_noise_scheduler", extra_globals=g)
        ddpm = extract(args.ref, "gradio_app.py", "PokemonGradioGenerator", method="ddpm_sample", extra_globals=g)
        fake = types.SimpleNamespace(config={"model": {"latent_dim": 8}}, device="cpu", num_timesteps=1000, beta_start=0.0001,
                                     beta_end=0.02, use_diffusers=False, unet=u)
        setup(fake)
        gt = O.gradio_tables()
        for n in ("betas", "alphas", "alphas_cumprod"):
            assert torch.equal(getattr(fake, n), gt[n]), n
        with torch.no_grad():
            x_ref = ddpm(fake, text, STEPS_GRADIO, init)
        x_start = init if init is not None else hashgen.uniform((1, 8, 27, 27), cases.INPUT_SEED, hashgen.name_id(tag + ".xT")) * math.sqrt(3.0)
        trace = []
        with torch.no_grad():
            x_or = O.gradio_ddpm_sample(eps_or, gt, text, x_start, noise_for(tag), STEPS_GRADIO, trace)
        report.append(f"gradio ddpm_sample ({tag}, {STEPS_GRADIO} steps): oracle vs reference max rel {maxrel(x_or, x_ref):.2e}")
        assert maxrel(x_or, x_ref) < 1e-4
        out[tag + "_final"] = x_ref.numpy()
        for i in (0, STEPS_GRADIO // 2, STEPS_GRADIO - 1):
            out[f"{tag}_step{i}"] = trace[i].numpy()

    np.savez_compressed(os.path.join(args.out, "inference.npz"), **out)
    with open(os.path.join(args.out, "REPORT_inference.txt"), "w") as f:
        f.write("oracle/make_golden_inference.py — oracle restatement vs reference source, torch %s\n" % torch.__version__)
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()

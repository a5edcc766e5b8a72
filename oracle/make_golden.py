#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — generate tests/golden/*.npz FROM THE REFERENCE.

Runs only in the authoring container, where /root/reference exists.  It loads
the reference's own source by file path (src/models/unet.py imports only torch;
NoiseScheduler and ddpm_sample are AST-extracted from modules whose top-level
imports need packages that are not installed), feeds it inputs and weights from
oracle/hashgen.py, checks oracle/unet_oracle.py against it, and writes OUTPUTS
ONLY as small fixtures.  No reference source, bytecode or pickled module is
written anywhere.

    python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import ast
import importlib.util
import math
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, hashgen, unet_oracle as O  # noqa: E402


def load_ref_unet(ref):
    spec = importlib.util.spec_from_file_location("ref_unet", os.path.join(ref, "src/models/unet.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def extract(ref, relpath, class_name, method=None, extra_globals=None):
    """AST-extract a class (or one method of it) from a reference module whose
    imports are unavailable, and exec it with only torch in scope."""
    src = open(os.path.join(ref, relpath)).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            if method is None:
                mod = ast.Module(body=[node], type_ignores=[])
            else:
                fn = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name == method][0]
                mod = ast.Module(body=[fn], type_ignores=[])
            g = {"torch": torch}
            g.update(extra_globals or {})
            exec(compile(mod, relpath, "exec"), g)
            return g[method or class_name]
    raise KeyError(class_name)


def digest(t, max_elems=4096):
    """Small pin of a big tensor: L2 norm, sum, and a strided sample."""
    f = t.detach().reshape(-1).double()
    stride = max(1, math.ceil(f.numel() / max_elems))
    return np.array([float(f.norm()), float(f.sum()), float(stride)]), f[::stride].float().numpy()


def maxrel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    R = load_ref_unet(args.ref)
    report = []

    # ---------------- a-1 schedule tables ---------------------------------
    NS = extract(args.ref, "src/training/improved_diffusion_trainer.py", "NoiseScheduler")
    NSlin = extract(args.ref, "src/training/diffusion_trainer.py", "NoiseScheduler")
    ns, nl = NS(), NSlin()
    names = ["betas", "alphas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"]
    sched = {}
    tc, tl = O.cosine_clipped_tables(), O.linear_tables()
    for n in names:
        sched["cos_" + n] = getattr(ns, n).numpy()
        sched["lin_" + n] = getattr(nl, n).numpy()
        assert torch.equal(getattr(ns, n), tc[n]), n       # bit-exact restatement
        assert torch.equal(getattr(nl, n), tl[n]), n
    ns2 = NS(num_timesteps=250, beta_start=0.0005, beta_end=0.03)
    t2 = O.cosine_clipped_tables(250, 0.0005, 0.03)
    for n in names:
        sched["cos250_" + n] = getattr(ns2, n).numpy()
        assert torch.equal(getattr(ns2, n), t2[n]), n
    np.savez_compressed(os.path.join(args.out, "schedule.npz"), **sched)
    report.append("schedule: oracle == reference bit-exact (cosine T=1000, T=250, linear)")

    # ---------------- a-2 add_noise ----------------------------------------
    x0 = hashgen.uniform((6, 8, 9, 9), cases.INPUT_SEED, hashgen.name_id("an.x0")) * 3.5
    nz = hashgen.uniform((6, 8, 9, 9), cases.INPUT_SEED, hashgen.name_id("an.noise")) * 2.0
    tt = torch.tensor([0, 500, 999, 37, 250, 998], dtype=torch.int64)
    x0c = torch.clamp(x0, -3.0, 3.0)                       # improved_diffusion_trainer.py:363
    out = ns.add_noise(x0c, nz, tt)
    assert torch.equal(out, O.add_noise(x0c, nz, tt, tc))
    out_lin = nl.add_noise(x0c, nz, tt)
    # NaN fallback path (:61-63)
    nzbad = nz.clone()
    nzbad[2, 3, 4, 5] = float("inf")
    out_bad = ns.add_noise(x0c, nzbad, tt)
    assert torch.equal(out_bad, O.add_noise(x0c, nzbad, tt, tc))
    np.savez_compressed(os.path.join(args.out, "add_noise.npz"), t=tt.numpy(), out=out.numpy(),
                        out_lin=out_lin.numpy(), out_fallback=out_bad.numpy())
    report.append("add_noise: oracle == reference bit-exact (incl. NaN/Inf fallback)")

    # ---------------- a-3 TimestepEmbedding --------------------------------
    blocks = {}
    te = R.TimestepEmbedding(128).eval()
    shapes = {k: tuple(v.shape) for k, v in te.state_dict().items()}
    sd = hashgen.fill_unet_state({"time_embed." + k: s for k, s in shapes.items()}, cases.WEIGHT_SEED, "stress")
    te.load_state_dict({k[len("time_embed."):]: v for k, v in sd.items()})
    tt = torch.tensor(cases.TIME_EMBED_T, dtype=torch.int64)
    ref_out = te(tt)
    ora = O.timestep_embedding(tt, sd)
    report.append(f"time_embed: max rel {maxrel(ora, ref_out):.2e}")
    assert maxrel(ora, ref_out) < 1e-5
    blocks["time_embed_out"] = ref_out.detach().numpy()

    # ---------------- a-4 ResBlock ------------------------------------------
    for name, cin, cout, hw, b in cases.RESBLOCK_CASES:
        m = R.ResBlock(cin, cout, 128, 256).eval()
        shapes = {"rb." + k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = hashgen.fill_unet_state(shapes, cases.WEIGHT_SEED, "stress")
        m.load_state_dict({k[3:]: v for k, v in sd.items()})
        x = (hashgen.uniform((b, cin, hw, hw), cases.INPUT_SEED, hashgen.name_id(name + ".x")) * 1.7).requires_grad_(True)
        temb = (hashgen.uniform((b, 128), cases.INPUT_SEED, hashgen.name_id(name + ".temb"))).requires_grad_(True)
        pooled = (hashgen.uniform((b, 256), cases.INPUT_SEED, hashgen.name_id(name + ".pooled"))).requires_grad_(True)
        gout = hashgen.uniform((b, cout, hw, hw), cases.INPUT_SEED, hashgen.name_id(name + ".gout"))
        y = m(x, temb, pooled)
        (y * gout).sum().backward()
        # oracle check
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        x2 = x.detach().clone().requires_grad_(True)
        t2_ = temb.detach().clone().requires_grad_(True)
        p2 = pooled.detach().clone().requires_grad_(True)
        y2 = O.resblock(x2, t2_, p2, leaves, "rb.")
        (y2 * gout).sum().backward()
        e = [maxrel(y2, y), maxrel(x2.grad, x.grad), maxrel(t2_.grad, temb.grad), maxrel(p2.grad, pooled.grad)]
        for k, p in m.named_parameters():
            e.append(maxrel(leaves["rb." + k].grad, p.grad))
        report.append(f"{name}: oracle vs reference max rel (fwd, grads) {max(e):.2e}")
        assert max(e) < 2e-5
        blocks[name + "_y"] = y.detach().numpy()
        blocks[name + "_dx"] = x.grad.numpy()
        blocks[name + "_dtemb"] = temb.grad.numpy()
        blocks[name + "_dpooled"] = pooled.grad.numpy()
        for k, p in m.named_parameters():
            d, s = digest(p.grad, 512)
            blocks[f"{name}_g_{k}_d"] = d
            blocks[f"{name}_g_{k}_s"] = s

    # ---------------- a-5 CrossAttentionBlock ------------------------------
    for name, ch, heads, hw, b, seq in cases.ATTN_CASES:
        m = R.CrossAttentionBlock(ch, 256, heads).eval()
        shapes = {"ab." + k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = hashgen.fill_unet_state(shapes, cases.WEIGHT_SEED, "stress")
        m.load_state_dict({k[3:]: v for k, v in sd.items()})
        x = (hashgen.uniform((b, ch, hw, hw), cases.INPUT_SEED, hashgen.name_id(name + ".x")) * 1.7).requires_grad_(True)
        text = hashgen.uniform((b, seq, 256), cases.INPUT_SEED, hashgen.name_id(name + ".text")) * 1.7
        gout = hashgen.uniform((b, ch, hw, hw), cases.INPUT_SEED, hashgen.name_id(name + ".gout"))
        y = m(x, text)
        (y * gout).sum().backward()
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        x2 = x.detach().clone().requires_grad_(True)
        y2 = O.cross_attention_block(x2, text, leaves, "ab.", heads)
        (y2 * gout).sum().backward()
        e = [maxrel(y2, y), maxrel(x2.grad, x.grad)]
        for k, p in m.named_parameters():
            e.append(maxrel(leaves["ab." + k].grad, p.grad))
        report.append(f"{name}: oracle vs reference max rel (fwd, grads) {max(e):.2e}")
        assert max(e) < 5e-5
        d, s = digest(y)
        blocks[name + "_y_d"], blocks[name + "_y_s"] = d, s
        d, s = digest(x.grad)
        blocks[name + "_dx_d"], blocks[name + "_dx_s"] = d, s
        for k, p in m.named_parameters():
            d, s = digest(p.grad, 512)
            blocks[f"{name}_g_{k}_d"] = d
            blocks[f"{name}_g_{k}_s"] = s
    np.savez_compressed(os.path.join(args.out, "blocks.npz"), **blocks)

    # ---------------- a-7 full-width U-Net ----------------------------------
    full = {}
    unets = {}
    for name, mode, b, ts, heads in cases.UNET_CASES + [cases.TRAIN_CASE]:
        key = (mode, heads)
        if key not in unets:
            t0 = time.time()
            u = R.UNet(8, 256, 128, heads).eval()
            shapes = {k: tuple(v.shape) for k, v in u.state_dict().items()}
            sd = hashgen.fill_unet_state(shapes, cases.WEIGHT_SEED, mode)
            u.load_state_dict(sd)
            unets[key] = (u, sd)
            print(f"built reference UNet {key} in {time.time() - t0:.1f}s", flush=True)
        u, sd = unets[key]
        x, t, text = hashgen.unet_inputs(b, cases.INPUT_SEED, t=ts)
        if name.startswith("unet_"):
            with torch.no_grad():
                y = u(x, t, text)
                yo = O.unet_forward(sd, x, t, text, heads)
            report.append(f"{name}: eps_hat mean|y|={float(y.abs().mean()):.4f} oracle vs reference max rel {maxrel(yo, y):.2e}")
            assert maxrel(yo, y) < 5e-5
            full[name + "_eps"] = y.numpy()
        else:
            # train-step body improved_diffusion_trainer.py:363-410, eval-mode (dropout off)
            noise = hashgen.uniform(x.shape, cases.INPUT_SEED, hashgen.name_id("train.noise")) * math.sqrt(3.0)
            lat = torch.clamp(x * 2.0, -3.0, 3.0)
            noisy = ns.add_noise(lat, noise, t)
            u.zero_grad()
            eps = u(noisy, t, text)
            loss = torch.nn.SmoothL1Loss(beta=0.1)(eps, noise)
            loss.backward()
            total = 0.0
            for p in u.parameters():                       # :399-404
                total += p.grad.data.norm(2).item() ** 2
            total = total ** 0.5
            r = O.train_step_grads(sd, x * 2.0, text, t, noise, tc, heads)
            e = [abs(r["loss"] - float(loss)) / float(loss), abs(r["grad_norm"] - total) / total, maxrel(r["eps_hat"], eps)]
            gn = []
            for k, p in u.named_parameters():
                e.append(maxrel(r["grads"][k], p.grad))
                gn.append(float(p.grad.double().norm()))
            report.append(f"{name}: loss={float(loss):.6f} |g|={total:.6f}  oracle vs reference max rel {max(e):.2e}")
            assert max(e) < 2e-4, e[:3]
            full[name + "_loss"] = np.array([float(loss)])
            full[name + "_grad_norm"] = np.array([total])
            full[name + "_eps"] = eps.detach().numpy()
            full[name + "_param_grad_norms"] = np.array(gn)
            for k in ["init_conv.weight", "final_conv.2.weight", "time_embed.time_mlp.0.weight",
                      "enc_block1.0.attn_block.cross_attn.in_proj_weight", "middle_block.res_block.conv1.weight",
                      "dec_block2.1.res_block.skip_conv.weight", "downsample2.weight", "upsample1.1.weight"]:
                d, s = digest(dict(u.named_parameters())[k].grad, 2048)
                full[f"{name}_g_{k}_d"], full[f"{name}_g_{k}_s"] = d, s
            clipped = torch.nn.utils.clip_grad_norm_(u.parameters(), max_norm=1.0)   # :410
            full[name + "_clip_total"] = np.array([float(clipped)])
            assert abs(O.clip_coef(total, 1.0) - min(1.0, 1.0 / (float(clipped) + 1e-6))) < 1e-6
    np.savez_compressed(os.path.join(args.out, "unet_full.npz"), **full)

    # ---------------- a-9 ddpm_sample (reference method, AST-extracted) -----
    name, mode, n, heads = cases.SAMPLE_CASE
    u, sd = unets[(mode, heads)]
    noise_log = []

    class TorchProxy:
        """torch with randn/randn_like replaced by the hash generator so the
        reference's own ddpm_sample body (improved_diffusion_trainer.py:508-569)
        runs on injected noise."""
        def __getattr__(self, a):
            return getattr(torch, a)

        def randn(self, shape, device=None):
            return hashgen.uniform(tuple(shape), cases.INPUT_SEED, hashgen.name_id("sample.xT")) * math.sqrt(3.0)

        def randn_like(self, x):
            i = len(noise_log)
            z = hashgen.uniform(tuple(x.shape), cases.INPUT_SEED, hashgen.name_id(f"sample.z{i}")) * math.sqrt(3.0)
            noise_log.append(i)
            return z

    ddpm = extract(args.ref, "src/training/improved_diffusion_trainer.py", "ImprovedDiffusionTrainer",
                   method="ddpm_sample", extra_globals={"torch": TorchProxy()})
    _, _, text = hashgen.unet_inputs(n, cases.INPUT_SEED)
    fake_self = types.SimpleNamespace(config={"model": {"latent_dim": 8}}, device="cpu", noise_scheduler=NS(), unet=u)
    t0 = time.time()
    x_ref = ddpm(fake_self, text, n, True)
    trace = []
    xT = hashgen.uniform((n, 8, 27, 27), cases.INPUT_SEED, hashgen.name_id("sample.xT")) * math.sqrt(3.0)

    def noise_fn(i, shape):
        return hashgen.uniform(tuple(shape), cases.INPUT_SEED, hashgen.name_id(f"sample.z{i}")) * math.sqrt(3.0)

    with torch.no_grad():
        x_or = O.ddpm_sample(lambda x, t, te: O.unet_forward(sd, x, t, te, heads), tc, text, xT, noise_fn, True, trace)
    report.append(f"{name}: 20-step fast sampler, oracle vs reference max rel {maxrel(x_or, x_ref):.2e} ({time.time() - t0:.0f}s)")
    assert maxrel(x_or, x_ref) < 1e-3
    smp = {"x_final": x_ref.numpy()}
    for i in (0, 4, 9, 14, 19):
        smp[f"x_step{i}"] = trace[i].numpy()
    np.savez_compressed(os.path.join(args.out, "sampler.npz"), **smp)

    with open(os.path.join(args.out, "REPORT.txt"), "w") as f:
        f.write("oracle/make_golden.py — oracle restatement vs reference source, torch %s\n" % torch.__version__)
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()

"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference hot path.

A functional (state_dict-driven) restatement, in plain torch CPU ops, of the
U-Net denoising train step of GabrieleConte/pokemon-sprite-generator.  Every
function cites the reference file:line it follows (paths relative to the
reference root).  It is pinned against the reference itself by
``oracle/make_golden.py`` (run in the authoring container, where the reference
source is importable by file path) and the fixtures in ``tests/golden``.

Parity status: the reference's own tests pin only I/O shapes, so
numeric parity is pinned by fixtures GENERATED from the reference source in the
authoring container (tests/golden/*.npz + oracle/make_golden.py), not by
reference-held golden vectors.

Never imported by the product path.
"""
import math

import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------
# a-1  NoiseScheduler tables  (src/training/improved_diffusion_trainer.py:25-48)
# ---------------------------------------------------------------------------


def cosine_clipped_tables(num_timesteps=1000, beta_start=0.0001, beta_end=0.02, s=0.008):
    """improved_diffusion_trainer.py:41-48 (cosine alpha-bar -> beta, clipped) then
    :29-39 (alphas, cumprod, sqrt tables clamped at 1e-8).  Same torch CPU ops in
    the same order, all fp32, so the tables are bit-identical to the reference's."""
    steps = num_timesteps + 1
    x = torch.linspace(0, num_timesteps, steps, dtype=torch.float32)
    ac = torch.cos(((x / num_timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    betas = torch.clip(betas, beta_start, beta_end).float()
    alphas = (1.0 - betas).float()
    alphas_cumprod = torch.cumprod(alphas, dim=0).float()
    sqrt_ac = torch.clamp(torch.sqrt(alphas_cumprod).float(), min=1e-8)
    sqrt_1m_ac = torch.clamp(torch.sqrt(1.0 - alphas_cumprod).float(), min=1e-8)
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_cumprod": alphas_cumprod,
        "sqrt_alphas_cumprod": sqrt_ac,
        "sqrt_one_minus_alphas_cumprod": sqrt_1m_ac,
    }


def linear_tables(num_timesteps=1000, beta_start=0.0001, beta_end=0.02):
    """Legacy linear-beta table, src/training/diffusion_trainer.py:29-35 (second
    known-answer for add_noise; the legacy trainer itself is out of scope)."""
    betas = torch.linspace(beta_start, beta_end, num_timesteps)
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_cumprod": alphas_cumprod,
        "sqrt_alphas_cumprod": torch.sqrt(alphas_cumprod),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - alphas_cumprod),
    }


# ---------------------------------------------------------------------------
# a-2  add_noise  (improved_diffusion_trainer.py:50-65, clamp :363)
# ---------------------------------------------------------------------------


def add_noise(x0, noise, t, tables):
    """improved_diffusion_trainer.py:55-65: a*x0 + b*noise with per-sample table
    gathers (two rounded multiplies, one rounded add); NaN/Inf -> x0 + 0.1*noise."""
    a = tables["sqrt_alphas_cumprod"][t].view(-1, 1, 1, 1)
    b = tables["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1, 1)
    noisy = a * x0 + b * noise
    if torch.isnan(noisy).any() or torch.isinf(noisy).any():
        return x0 + 0.1 * noise
    return noisy


# ---------------------------------------------------------------------------
# a-3  TimestepEmbedding  (src/models/unet.py:12-55)
# ---------------------------------------------------------------------------


def timestep_embedding(t, sd, prefix="time_embed."):
    """unet.py:47-53: t.float() * coeff, cat[sin, cos], Linear-SiLU-Linear-SiLU-Linear."""
    e = t.float().unsqueeze(-1) * sd[prefix + "emb_coeff"].unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=-1)
    h = F.silu(F.linear(e, sd[prefix + "time_mlp.0.weight"], sd[prefix + "time_mlp.0.bias"]))
    h = F.silu(F.linear(h, sd[prefix + "time_mlp.2.weight"], sd[prefix + "time_mlp.2.bias"]))
    return F.linear(h, sd[prefix + "time_mlp.4.weight"], sd[prefix + "time_mlp.4.bias"])


# ---------------------------------------------------------------------------
# a-4  ResBlock  (unet.py:58-132)
# ---------------------------------------------------------------------------


def _groups(c):
    """unet.py:70-76 / :151-153: largest g <= 32 dividing c."""
    g = min(32, c)
    while c % g != 0 and g > 1:
        g -= 1
    return max(1, g)


def resblock(x, temb, pooled, sd, p):
    """unet.py:112-132.  GN eps 1e-5; dropout p=0.0 is the identity."""
    cin = x.shape[1]
    cout = sd[p + "conv1.weight"].shape[0]
    h = F.silu(F.group_norm(x, _groups(cin), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5))
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = h + F.linear(temb, sd[p + "time_proj.weight"], sd[p + "time_proj.bias"])[:, :, None, None]
    h = h + F.linear(pooled, sd[p + "text_proj.weight"], sd[p + "text_proj.bias"])[:, :, None, None]
    h = F.silu(F.group_norm(h, _groups(cout), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5))
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if (p + "skip_conv.weight") in sd:
        skip = F.conv2d(x, sd[p + "skip_conv.weight"], sd[p + "skip_conv.bias"])
    else:
        skip = x
    return h + skip


# ---------------------------------------------------------------------------
# a-5  CrossAttentionBlock  (unet.py:135-260)
# ---------------------------------------------------------------------------


def mha(q_in, kv_in, w, b, wo, bo, num_heads):
    """nn.MultiheadAttention(batch_first=True) math path restated
    (torch/nn/functional.py multi_head_attention_forward): packed in-proj rows
    [0:E] -> q, [E:2E] -> k, [2E:3E] -> v; q scaled by 1/sqrt(d) BEFORE QK^T;
    softmax over keys; no mask; dropout off (eval); out-proj.  Called at
    unet.py:217 (self) and :235 (cross, key=value=text_proj(text))."""
    B, L, E = q_in.shape
    S = kv_in.shape[1]
    d = E // num_heads
    q = F.linear(q_in, w[:E], b[:E])
    k = F.linear(kv_in, w[E:2 * E], b[E:2 * E])
    v = F.linear(kv_in, w[2 * E:], b[2 * E:])
    q = q.view(B, L, num_heads, d).transpose(1, 2) * math.sqrt(1.0 / d)
    k = k.view(B, S, num_heads, d).transpose(1, 2)
    v = v.view(B, S, num_heads, d).transpose(1, 2)
    attn = torch.softmax(q @ k.transpose(-2, -1), dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, L, E)
    return F.linear(o, wo, bo)


def cross_attention_block(x, text, sd, p, num_heads):
    """unet.py:206-260 (eval mode: every Dropout is the identity).  GN eps 1e-6 on
    [B,C,L]; gates 0.7 / 0.8 / 0.6; no norm before the FFN; GELU is erf-form."""
    B, C, H, W = x.shape
    g = _groups(C)
    xf = x.view(B, C, H * W).permute(0, 2, 1)
    # self-attention :212-221
    xn = F.group_norm(xf.permute(0, 2, 1), g, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6).permute(0, 2, 1)
    a = mha(xn, xn, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"],
            sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], num_heads)
    xf = xf + a * 0.7
    # cross-attention :229-239
    xn = F.group_norm(xf.permute(0, 2, 1), g, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6).permute(0, 2, 1)
    tp = F.linear(text, sd[p + "text_proj.weight"], sd[p + "text_proj.bias"])
    a = mha(xn, tp, sd[p + "cross_attn.in_proj_weight"], sd[p + "cross_attn.in_proj_bias"],
            sd[p + "cross_attn.out_proj.weight"], sd[p + "cross_attn.out_proj.bias"], num_heads)
    xf = xf + a * 0.8
    # FFN :247-251
    h = F.gelu(F.linear(xf, sd[p + "ffn.0.weight"], sd[p + "ffn.0.bias"]))
    h = F.linear(h, sd[p + "ffn.3.weight"], sd[p + "ffn.3.bias"])
    xf = xf + h * 0.6
    return xf.permute(0, 2, 1).reshape(B, C, H, W)


def unet_block(x, temb, pooled, text, sd, p, num_heads):
    """unet.py:295-301."""
    x = resblock(x, temb, pooled, sd, p + "res_block.")
    if (p + "attn_block.norm1.weight") in sd:
        x = cross_attention_block(x, text, sd, p + "attn_block.", num_heads)
    return x


# ---------------------------------------------------------------------------
# a-7  UNet.forward  (unet.py:428-509)
# ---------------------------------------------------------------------------


def unet_forward(sd, noisy_latent, timesteps, text_emb, num_heads=8):
    """unet.py:442-509.  The same skip tensor feeds both decoder blocks of a
    level (:480-504); upsample = bilinear (align_corners=False) to the fixed
    sizes 7/14/27 then a 3x3 conv (:364-387)."""
    temb = timestep_embedding(timesteps, sd)
    pooled = text_emb.mean(dim=1)                      # AdaptiveAvgPool1d(1) :445
    x = F.conv2d(noisy_latent, sd["init_conv.weight"], sd["init_conv.bias"], padding=1)
    skips = []
    for lvl in range(4):
        if lvl > 0:
            x = F.conv2d(x, sd[f"downsample{lvl}.weight"], sd[f"downsample{lvl}.bias"], stride=2, padding=1)
        for i in range(2):
            x = unet_block(x, temb, pooled, text_emb, sd, f"enc_block{lvl}.{i}.", num_heads)
        skips.append(x)
    x = unet_block(x, temb, pooled, text_emb, sd, "middle_block.", num_heads)
    sizes = {3: (7, 7), 2: (14, 14), 1: (27, 27)}
    for lvl in (3, 2, 1, 0):
        skip = skips.pop()
        for i in range(2):
            x = torch.cat([x, skip], dim=1)
            x = unet_block(x, temb, pooled, text_emb, sd, f"dec_block{lvl}.{i}.", num_heads)
        if lvl > 0:
            x = F.interpolate(x, size=sizes[lvl], mode="bilinear", align_corners=False)
            x = F.conv2d(x, sd[f"upsample{lvl}.1.weight"], sd[f"upsample{lvl}.1.bias"], padding=1)
    x = F.silu(F.group_norm(x, 32, sd["final_conv.0.weight"], sd["final_conv.0.bias"], 1e-5))
    return F.conv2d(x, sd["final_conv.2.weight"], sd["final_conv.2.bias"], padding=1)


# ---------------------------------------------------------------------------
# a-8  train-step body  (improved_diffusion_trainer.py:363-413)
# ---------------------------------------------------------------------------


def smooth_l1(pred, target, beta=0.1):
    """nn.SmoothL1Loss(beta=0.1), mean reduction (:300, :388)."""
    d = (pred - target).abs()
    return torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta).mean()


def clip_coef(total_norm, max_norm):
    """torch.nn.utils.clip_grad_norm_ (:410): coef = max_norm/(norm+1e-6) clamped to 1."""
    return min(1.0, max_norm / (total_norm + 1e-6))


def adamw_update(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
    """torch.optim.AdamW single-tensor math (:277-283, :412), eps=1e-6 in the trainer.
    Returns (p, m, v) updated; step is 1-based."""
    p = p * (1 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def train_step_grads(sd, latents, text_emb, t, noise, tables, num_heads=8, param_keys=None):
    """One step body :363-410 with dropout disabled: clamp, add_noise, forward,
    SmoothL1, backward, global grad norm.  Returns dict(loss, eps_hat, grads,
    grad_norm)."""
    keys = param_keys or [k for k in sd if not k.endswith("emb_coeff")]
    leaves = {k: (sd[k].detach().clone().requires_grad_(True) if k in keys else sd[k]) for k in sd}
    x0 = torch.clamp(latents, -3.0, 3.0)
    noisy = add_noise(x0, noise, t, tables)
    eps_hat = unet_forward(leaves, noisy, t, text_emb, num_heads)
    loss = smooth_l1(eps_hat, noise, 0.1)
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys])
    total = 0.0
    for g in grads:                       # improved_diffusion_trainer.py:399-404 (fp32 per-tensor norms)
        total += g.norm(2).item() ** 2
    total = total ** 0.5
    return {
        "loss": float(loss.detach()),
        "eps_hat": eps_hat.detach(),
        "noisy": noisy,
        "grads": dict(zip(keys, grads)),
        "grad_norm": total,
    }


# ---------------------------------------------------------------------------
# a-9  ddpm_sample  (improved_diffusion_trainer.py:508-569)
# ---------------------------------------------------------------------------


def ddpm_timesteps(num_timesteps=1000, fast_sampling=True):
    """:528-534."""
    ts = list(range(0, num_timesteps, 50)) if fast_sampling else list(range(num_timesteps))
    return list(reversed(ts))


def ddpm_step_coeffs(tables, t):
    """:543-555, :562: c1 = 1/sqrt(alpha_t), c2 = beta_t/sqrt(1-alphabar_t),
    sigma = sqrt(beta_t) (NOT the posterior variance), as 0-d fp32 tensors."""
    alpha_t = tables["alphas"][t]
    ac_t = tables["alphas_cumprod"][t]
    beta_t = tables["betas"][t]
    c1 = 1.0 / torch.sqrt(alpha_t)
    c2 = beta_t / torch.sqrt(1 - ac_t)
    return c1, c2, torch.sqrt(beta_t)


def ddpm_sample(eps_fn, tables, text_emb, x_T, noise_fn, fast_sampling=True, trace=None):
    """:534-569.  eps_fn(x, t_vec, text) -> eps_hat; noise_fn(step_index, shape) ->
    z (injected so traces are reproducible).  In fast mode t%50==0 always holds,
    so noise is added whenever t > 0 in both modes (:560-567)."""
    x = x_T
    n = x.shape[0]
    num_t = tables["betas"].shape[0]
    for i, t in enumerate(ddpm_timesteps(num_t, fast_sampling)):
        tv = torch.full((n,), t, dtype=torch.long)
        eps = eps_fn(x, tv, text_emb)
        c1, c2, sigma = ddpm_step_coeffs(tables, t)
        x = c1 * (x - c2 * eps)
        if t > 0:
            x = x + sigma * noise_fn(i, x.shape)
        if trace is not None:
            trace.append(x.clone())
    return x


# ---------------------------------------------------------------------------
# f-3  inference consumers of the trained U-Net
# ---------------------------------------------------------------------------


def final_linear_tables(num_timesteps=1000, beta_start=0.0001, beta_end=0.02):
    """src/training/final_trainer.py:22-40 - the stage-3 NoiseScheduler: LINEAR betas plus sqrt_recip_alphas and the
    clamped posterior variance."""
    betas = torch.linspace(beta_start, beta_end, num_timesteps)
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    posterior_variance = betas * (1.0 - torch.cat([torch.tensor([1.0]), alphas_cumprod[:-1]])) / (1.0 - alphas_cumprod)
    return {
        "betas": betas, "alphas": alphas, "alphas_cumprod": alphas_cumprod,
        "sqrt_alphas_cumprod": torch.sqrt(alphas_cumprod),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - alphas_cumprod),
        "sqrt_recip_alphas": torch.sqrt(1.0 / alphas),
        "posterior_variance": torch.clamp(posterior_variance, min=1e-20),
    }


def final_add_noise(x0, noise, t, tables):
    """final_trainer.py:42-50 (no clamp, no fallback)."""
    a = tables["sqrt_alphas_cumprod"][t].view(-1, 1, 1, 1)
    b = tables["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1, 1)
    return a * x0 + b * noise


def sample_previous_timestep(x_t, predicted_noise, timestep, tables, noise=None):
    """final_trainer.py:52-71: mean = sqrt(1/alpha_t) * (x_t - beta_t * eps / sqrt(1 - abar_t)); for t > 0 plus
    sqrt(posterior_variance_t) * noise (`noise` is what the reference draws with randn_like)."""
    sra = tables["sqrt_recip_alphas"][timestep]
    beta = tables["betas"][timestep]
    somac = tables["sqrt_one_minus_alphas_cumprod"][timestep]
    mean = sra * (x_t - beta * predicted_noise / somac)
    if timestep > 0:
        return mean + torch.sqrt(tables["posterior_variance"][timestep]) * noise
    return mean


def final_generate_latents(eps_fn, tables, text_emb, x_T, noise_fn, num_inference_steps=50, trace=None):
    """final_trainer.py:186-204 - the latent loop of FinalPokemonGenerator.forward(mode='generate'): timestep
    T-1 - i*step_size clamped at 0; sample_previous_timestep for t > 0, `latent - predicted_noise` at t == 0.
    noise_fn(i, shape) supplies the i-th randn_like draw."""
    num_t = tables["betas"].shape[0]
    n = x_T.shape[0]
    latent = x_T
    step_size = max(1, num_t // num_inference_steps)
    draws = 0
    for i in range(num_inference_steps):
        timestep = max(0, num_t - 1 - i * step_size)
        tv = torch.full((n,), timestep, dtype=torch.long)
        eps = eps_fn(latent, tv, text_emb)
        if timestep > 0:
            latent = sample_previous_timestep(latent, eps, timestep, tables, noise_fn(draws, latent.shape))
            draws += 1
        else:
            latent = latent - eps
        if trace is not None:
            trace.append(latent.clone())
    return latent


def gradio_tables(num_timesteps=1000, beta_start=0.0001, beta_end=0.02):
    """gradio_app.py:279-288 (linear betas)."""
    betas = torch.linspace(beta_start, beta_end, num_timesteps)
    alphas = 1.0 - betas
    return {"betas": betas, "alphas": alphas, "alphas_cumprod": torch.cumprod(alphas, dim=0)}


def gradio_ddpm_sample(eps_fn, tables, text_emb, x_init, noise_fn, num_inference_steps=50, trace=None):
    """gradio_app.py:323-361 - the demo's own sampler: timesteps = linspace(T-1, 0, n) truncated to long; per step
    latent = (latent - (1-alpha_t)/sqrt(1-abar_t) * eps) / sqrt(alpha_t), then (not on the last step, and only if the
    NEXT timestep is > 0) re-noised: sqrt(alpha_next) * latent + sqrt(1 - alpha_next) * noise."""
    num_t = tables["betas"].shape[0]
    n = x_init.shape[0]
    latent = x_init.clone()
    timesteps = torch.linspace(num_t - 1, 0, num_inference_steps, dtype=torch.long)
    alphas, ac = tables["alphas"], tables["alphas_cumprod"]
    draws = 0
    for i, t in enumerate(timesteps):
        tv = torch.full((n,), t.item(), dtype=torch.long)
        eps = eps_fn(latent, tv, text_emb)
        if i < len(timesteps) - 1:
            next_t = timesteps[i + 1]
            alpha_t, alpha_next = alphas[t], alphas[next_t]
            latent = (latent - (1 - alpha_t) / torch.sqrt(1 - ac[t]) * eps) / torch.sqrt(alpha_t)
            if next_t > 0:
                noise = noise_fn(draws, latent.shape)
                draws += 1
                latent = torch.sqrt(alpha_next) * latent + torch.sqrt(1 - alpha_next) * noise
        else:
            latent = (latent - (1 - alphas[t]) / torch.sqrt(1 - ac[t]) * eps) / torch.sqrt(alphas[t])
        if trace is not None:
            trace.append(latent.clone())
    return latent

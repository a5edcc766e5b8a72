"""Inference consumers of the trained U-Net, on the accelerated forward.

Mirrors of the two other samplers in the reference that load a stage-2 checkpoint's 'unet_state_dict':

  * `LinearNoiseScheduler` — stage 3's NoiseScheduler (src/training/final_trainer.py:19-81): linear betas,
    `sqrt_recip_alphas`, clamped `posterior_variance`, `add_noise`, `sample_previous_timestep`.
  * `LatentGenerator.forward` — the latent loop of `FinalPokemonGenerator.forward(mode='generate')`
    (final_trainer.py:165-204): strided timesteps T-1-i*step, posterior-variance noise, `latent - eps` at t == 0.
    The BERT text encoder and the VAE decoder either side of it are out of scope: the class takes text EMBEDDINGS and
    returns latents, and applies `vae_decoder(latent, text_emb)` when one is given.
  * `gradio_ddpm_sample` — `PokemonGradioGenerator.setup_noise_scheduler` + `ddpm_sample` (gradio_app.py:279-361),
    text-only or from an initial latent.

Every schedule scalar is computed on the host with the reference's own torch expressions; the per-element update runs in
`psg_sampler_update_f32` with the reference's operation order (bit-exact against the CPU path for a given eps).
"""
import math
from typing import Callable, Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _lib_for(t):
    if not t.is_cuda:
        raise _lib.PsgError("the samplers need GPU tensors (HIP path only; no CPU fallback)")
    return _lib.init(t.device.index if t.device.index is not None else torch.cuda.current_device())


def _f32(v):
    return np.float32(float(v))


def _sqrt32(v):
    """Correctly rounded fp32 square root of an fp32 scalar (double sqrt, one rounding).  torch's vectorised CPU sqrt is
    not the same function on every host CPU (last bit); the per-step scalars must not depend on the host."""
    return np.float32(math.sqrt(float(v)))


def _update(x, eps, z, mode, c0=0.0, c1=1.0, c2=0.0, c3=0.0):
    lib = _lib_for(x)
    eps = eps.detach().contiguous().float()
    if z is not None:
        z = z.detach().to(device=x.device, dtype=torch.float32).contiguous()
    check(lib.psg_sampler_update_f32(ptr(x), ptr(eps), ptr(z), int(mode), float(c0), float(c1), float(c2), float(c3), x.numel(),
                                     stream_ptr()), "psg_sampler_update_f32")
    return x


class LinearNoiseScheduler:
    """final_trainer.py:19-81.  Tables live on the host (they are 4 KB of scalars); tensors passed in stay on the GPU."""

    def __init__(self, num_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02):
        self.num_timesteps = num_timesteps
        self.betas = torch.linspace(beta_start, beta_end, num_timesteps)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas = torch.sqrt(1.0 / self.alphas)
        posterior_variance = self.betas * (1.0 - torch.cat([torch.tensor([1.0]), self.alphas_cumprod[:-1]])) / (1.0 - self.alphas_cumprod)
        self.posterior_variance = torch.clamp(posterior_variance, min=1e-20)
        self._dev = {}

    def to(self, device):
        """Kept for signature parity (the reference moves its tables; here only the two gather tables of add_noise
        are mirrored on the device, lazily)."""
        return self

    def _dev_tables(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (self.sqrt_alphas_cumprod.to(device).contiguous(), self.sqrt_one_minus_alphas_cumprod.to(device).contiguous())
        return self._dev[key]

    def add_noise(self, x_0: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """:42-50 - no clamp and no NaN fallback in this scheduler."""
        lib = _lib_for(x_0)
        x0, nz = x_0.detach().contiguous().float(), noise.detach().contiguous().float()
        t = timesteps.detach().to(device=x_0.device, dtype=torch.int64).contiguous()
        a, b = self._dev_tables(x_0.device)
        out = torch.empty_like(x0)
        flag = torch.zeros(1, dtype=torch.int32, device=x_0.device)
        B = x0.shape[0]
        if B:
            check(lib.psg_noise_add_f32(ptr(x0), ptr(nz), ptr(t), ptr(a), ptr(b), ptr(out), ptr(flag), B, x0.numel() // B,
                                        self.num_timesteps, 0, stream_ptr()), "psg_noise_add_f32")
        return out

    def sample_previous_timestep(self, x_t: torch.Tensor, predicted_noise: torch.Tensor, timestep: int,
                                 noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """:52-71.  `noise` (default randn_like) is the draw the reference makes for timestep > 0."""
        t = int(timestep)
        x = x_t.detach().float().contiguous().clone()
        z = None
        if t > 0:
            z = noise if noise is not None else torch.randn_like(x)
        sv = float(_sqrt32(self.posterior_variance[t])) if t > 0 else 0.0
        return _update(x, predicted_noise, z, 1, float(self.sqrt_recip_alphas[t]), float(self.betas[t]),
                       float(self.sqrt_one_minus_alphas_cumprod[t]), sv)


class LatentGenerator:
    """The denoising loop of FinalPokemonGenerator.forward(mode='generate') (final_trainer.py:186-204) around a
    `pokemon_sprite_generator_amd.UNet` (weights: a stage-2 checkpoint's 'unet_state_dict', loaded as is)."""

    def __init__(self, unet, noise_scheduler: Optional[LinearNoiseScheduler] = None, latent_dim: int = 8,
                 vae_decoder: Optional[Callable] = None):
        self.unet, self.latent_dim, self.vae_decoder = unet, latent_dim, vae_decoder
        self.noise_scheduler = noise_scheduler or LinearNoiseScheduler()

    @torch.no_grad()
    def forward(self, text_emb: torch.Tensor, num_inference_steps: int = 50, noise_fn=None, trace=None):
        """text_emb [B,S,text_dim] (what `self.text_encoder(text_list)` returns in the reference).
        noise_fn(i, shape): i = -1 the initial latent, i >= 0 the i-th randn_like draw (tests inject them)."""
        self.unet.eval()
        dev = next(self.unet.parameters()).device
        text_emb = text_emb.to(dev)
        B = text_emb.shape[0]
        rnd = noise_fn if noise_fn is not None else (lambda i, shape: torch.randn(shape, device=dev))
        latent = rnd(-1, (B, self.latent_dim, 27, 27)).to(device=dev, dtype=torch.float32).contiguous().clone()
        sch = self.noise_scheduler
        step_size = max(1, sch.num_timesteps // num_inference_steps)
        draws = 0
        for i in range(num_inference_steps):
            timestep = max(0, sch.num_timesteps - 1 - i * step_size)
            tv = torch.full((B,), timestep, device=dev, dtype=torch.long)
            eps = self.unet(latent, tv, text_emb)
            if timestep > 0:
                latent = sch.sample_previous_timestep(latent, eps, timestep, noise=rnd(draws, tuple(latent.shape)))
                draws += 1
            else:
                latent = _update(latent.clone(), eps, None, 2)
            if trace is not None:
                trace.append(latent.clone())
        return self.vae_decoder(latent, text_emb) if self.vae_decoder is not None else latent

    __call__ = forward


@torch.no_grad()
def gradio_ddpm_sample(unet, text_emb: torch.Tensor, num_inference_steps: int = 50, initial_latent: Optional[torch.Tensor] = None,
                       num_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02, latent_dim: int = 8,
                       noise_fn=None, trace=None) -> torch.Tensor:
    """PokemonGradioGenerator.ddpm_sample (gradio_app.py:297-361) with its own linear schedule (:279-288)."""
    unet.eval()
    dev = next(unet.parameters()).device
    betas = torch.linspace(beta_start, beta_end, num_timesteps)
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    text_emb = text_emb.to(dev)
    B = text_emb.shape[0]
    rnd = noise_fn if noise_fn is not None else (lambda i, shape: torch.randn(shape, device=dev))
    if initial_latent is None:
        latent = rnd(-1, (B, latent_dim, 27, 27)).to(device=dev, dtype=torch.float32).contiguous().clone()
    else:
        latent = initial_latent.to(device=dev, dtype=torch.float32).contiguous().clone()
    timesteps = torch.linspace(num_timesteps - 1, 0, num_inference_steps, dtype=torch.long)
    draws = 0
    for i, t in enumerate(timesteps):
        tv = torch.full((B,), t.item(), dtype=torch.long, device=dev)
        eps = unet(latent, tv, text_emb)
        one = np.float32(1.0)                                   # fp32 scalar arithmetic, as the reference's 0-d tensors
        a_t, ac_t = _f32(alphas[t]), _f32(alphas_cumprod[t])
        k0 = float((one - a_t) / _sqrt32(one - ac_t))
        k1 = float(_sqrt32(a_t))
        z, c2, c3 = None, 0.0, 0.0
        if i < len(timesteps) - 1:
            next_t = timesteps[i + 1]
            if next_t > 0:
                z = rnd(draws, tuple(latent.shape))
                draws += 1
                a_n = _f32(alphas[next_t])
                c2, c3 = float(_sqrt32(a_n)), float(_sqrt32(one - a_n))
        _update(latent, eps, z, 3, k0, k1, c2, c3)
        if trace is not None:
            trace.append(latent.clone())
    return latent

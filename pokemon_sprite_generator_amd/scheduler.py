"""NoiseScheduler mirror (reference: src/training/improved_diffusion_trainer.py:22-74).

The five schedule tables are built on the HOST with the same torch CPU ops in the
same order as the reference (bit-identical fp32 tables); the
per-sample gather + `a*x0 + b*noise` runs in psg_noise_add_f32 with separately
rounded multiplies and add, so results are bit-identical to the CPU path.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

def _canon(device):
    """torch.device with an explicit index for GPUs ('cuda' -> 'cuda:<current>')."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


_TABLES = ("betas", "alphas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod")


class NoiseScheduler:
    """Cosine alpha-bar schedule converted to betas and clipped to [beta_start, beta_end]."""

    def __init__(self, num_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02):
        self.num_timesteps = num_timesteps
        self.betas = self._cosine_beta_schedule(num_timesteps, beta_start, beta_end).float()
        self.alphas = (1.0 - self.betas).float()
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0).float()
        self.sqrt_alphas_cumprod = torch.clamp(torch.sqrt(self.alphas_cumprod).float(), min=1e-8)
        self.sqrt_one_minus_alphas_cumprod = torch.clamp(torch.sqrt(1.0 - self.alphas_cumprod).float(), min=1e-8)
        self._flag = None
        self._step_tables = None

    @staticmethod
    def _cosine_beta_schedule(timesteps, beta_start, beta_end, s=0.008):
        x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float32)
        acp = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
        acp = acp / acp[0]
        return torch.clip(1 - (acp[1:] / acp[:-1]), beta_start, beta_end)

    def to(self, device):
        """Move the tables to `device` (idempotent, unlike the reference's per-call re-upload)."""
        device = _canon(device)
        if self.betas.device != device:
            for n in _TABLES:
                setattr(self, n, getattr(self, n).to(device, dtype=torch.float32))
            self._flag = None
            self._step_tables = None
        return self

    # -- device-side NaN/Inf flag (bit 0: non-finite output, bit 1: timestep out of range) ------------
    def nan_flag(self, device):
        device = _canon(device)
        if self._flag is None or self._flag.device != device:
            self._flag = torch.zeros(1, dtype=torch.int32, device=device)
        return self._flag

    def add_noise(self, x_0: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor, clamp: bool = False,
                  flag: torch.Tensor = None) -> torch.Tensor:
        """noisy = sqrt(abar_t)*x0 + sqrt(1-abar_t)*noise; non-finite result -> x0 + 0.1*noise (:61-63),
        decided on the device (no host sync).  `clamp=True` fuses the trainer's clamp(latent,-3,3) (:363)."""
        if not x_0.is_cuda:
            raise _lib.PsgError("NoiseScheduler.add_noise needs GPU tensors (HIP path only; no CPU fallback)")
        lib = _lib.init(x_0.device.index if x_0.device.index is not None else torch.cuda.current_device())
        self.to(x_0.device)
        x0 = x_0.detach().contiguous().float()
        nz = noise.detach().contiguous().float()
        t = timesteps.detach().to(device=x_0.device, dtype=torch.int64).contiguous()
        B = x0.shape[0]
        chw = x0.numel() // max(B, 1)
        if t.numel() != B or nz.shape != x0.shape:
            raise ValueError(f"add_noise: shapes x0 {tuple(x0.shape)}, noise {tuple(nz.shape)}, t {tuple(t.shape)}")
        out = torch.empty_like(x0)
        if B == 0:
            return out
        fl = flag if flag is not None else self.nan_flag(x_0.device)
        if flag is None:
            fl.zero_()
        check(lib.psg_noise_add_f32(ptr(x0), ptr(nz), ptr(t), ptr(self.sqrt_alphas_cumprod), ptr(self.sqrt_one_minus_alphas_cumprod),
                                    ptr(out), ptr(fl), B, chw, self.num_timesteps, int(clamp), stream_ptr()), "psg_noise_add_f32")
        check(lib.psg_noise_fallback_f32(ptr(x0), ptr(nz), ptr(out), ptr(fl), x0.numel(), int(clamp), stream_ptr()),
              "psg_noise_fallback_f32")
        return out

    # -- per-timestep sampler scalars (ddpm_sample :543-562), same torch ops as the reference -----------
    def step_tables(self, device):
        """c1 = 1/sqrt(alpha_t), c2 = beta_t/sqrt(1-abar_t), sigma = sqrt(beta_t) for every t (fp32 [T])."""
        device = _canon(device)
        if self._step_tables is None or self._step_tables[0].device != device:
            a, ac, b = self.alphas.cpu(), self.alphas_cumprod.cpu(), self.betas.cpu()
            c1 = 1.0 / torch.sqrt(a)
            c2 = b / torch.sqrt(1 - ac)
            sg = torch.sqrt(b)
            self._step_tables = tuple(t.to(device).contiguous() for t in (c1, c2, sg))
        return self._step_tables

"""Gradient arena + fused clip/AdamW (reference: improved_diffusion_trainer.py:277-283, 399-413).

GradArena keeps every parameter gradient in ONE flat fp32 buffer (p.grad are
persistent views) and registers each view as a GradSink: the wgrad / column-sum
/ GroupNorm-backward kernels write gradients straight into it.  The global L2
norm is then one deterministic reduction (psg_sumsq_f32) instead of 478
`.item()` syncs, clip + AdamW is fused (psg_adamw_f32) and the data-parallel
all-reduce runs on large flat slices.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .ops import GradSink, WeightCache


class GradArena:
    def __init__(self, params, on_ready=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradArena: no trainable parameters")
        dev = self.params[0].device
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4          # keep every view 16-byte aligned
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.normsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.on_ready = on_ready
        self.views = []
        GradSink.unregister_all()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            v = self.flat[o:o + p.numel()].view_as(p)
            self.views.append(v)
            p.grad = v
            GradSink.register(p, v, i, self._ready)

    def _ready(self, index):
        if self.on_ready is not None:
            self.on_ready(index)

    def zero(self):
        """Start of a step: nothing is memset — every sink is overwritten by its first gradient kernel."""
        GradSink.begin_step()
        for p, v in zip(self.params, self.views):       # re-attach if someone set grads to None
            if p.grad is not v:
                p.grad = v

    def finalize(self):
        """After backward: parameters that received no gradient this step get an explicit zero."""
        for e in GradSink.unwritten():
            e.view.zero_()
            GradSink.done(e)

    def grad_norm_sq(self):
        """Device scalar sum(g^2) over all parameters (padding is zero)."""
        lib = _lib.init(self.flat.device.index)
        ws = _lib.workspace(lib.psg_reduce_workspace_bytes(), self.flat.device)
        check(lib.psg_sumsq_f32(ptr(self.flat), self.numel, ptr(self.normsq), 0, ptr(ws), stream_ptr()), "psg_sumsq_f32")
        return self.normsq


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt-ratio)
    with gradient clipping folded in.  state_dict layout matches torch.optim.AdamW
    ('step', 'exp_avg', 'exp_avg_sq' per parameter), so reference checkpoints interchange."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, normsq=None, max_norm=0.0, skip_flag=None, grad_scale=1.0):
        lib = None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if lib is None:
                    lib = _lib.init(p.device.index)
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                check(lib.psg_adamw_f32(ptr(p), ptr(g), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), p.numel(), float(group["lr"]),
                                        float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), int(st["step"]),
                                        ptr(normsq) if normsq is not None else None, float(max_norm),
                                        ptr(skip_flag) if skip_flag is not None else None, stream_ptr()), "psg_adamw_f32")
        WeightCache.invalidate()     # parameters changed through raw pointers

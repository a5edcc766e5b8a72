"""Shared helpers for the parity tests (test infrastructure)."""
import math

import numpy as np
import torch

from oracle import hashgen

SEED = 4242


def h(shape, name, scale=1.0, seed=SEED):
    return hashgen.uniform(tuple(shape), seed, hashgen.name_id(name)) * scale


def maxrel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def digest(t, max_elems=4096):
    f = t.detach().reshape(-1).double().cpu()
    stride = max(1, math.ceil(f.numel() / max_elems))
    return np.array([float(f.norm()), float(f.sum()), float(stride)]), f[::stride].float().numpy()


def check_digest(t, d, s, tol, what=""):
    """Compare tensor t with a stored (norm,sum,stride) + strided sample digest."""
    dd, ss = digest(t, max_elems=max(1, int(math.ceil(t.numel() / d[2]))) if d[2] > 1 else t.numel())
    f = t.detach().reshape(-1).double().cpu()
    sample = f[::int(d[2])].float().numpy()
    assert sample.shape == s.shape, f"{what}: sample shape {sample.shape} vs {s.shape}"
    scale = np.abs(s).max() + 1e-30
    err = np.abs(sample - s).max() / scale
    assert err < tol, f"{what}: sample max rel err {err:.3e} >= {tol}"
    nerr = abs(float(f.norm()) - d[0]) / (d[0] + 1e-30)
    assert nerr < tol, f"{what}: norm rel err {nerr:.3e} >= {tol}"


def to_cl(x, dtype=torch.float32):
    """NCHW -> channels-last [B,H,W,C] contiguous."""
    return x.permute(0, 2, 3, 1).contiguous().to(dtype)


def from_cl(x):
    return x.float().permute(0, 3, 1, 2).contiguous()


TOL = {torch.float32: 2e-4, torch.bfloat16: 3e-2}

"""TEST INFRASTRUCTURE ONLY — seeded integer-hash tensor generator.

Deterministic, platform- and torch-version-independent pseudo-random tensors, so
that the 640 M U-Net weights (2.56 GB) and every test input can be regenerated
bit-identically on the GPU box and never need committing (SURVEY.md §7 step 0,
§8c "procedurally generated weights").

value(seed, tid, i) = murmur3-fmix32(i*0x9E3779B1 + tid*0x85EBCA77 + seed*0xC2B2AE3D)
mapped to a float32 uniform in [-1, 1) with 24 mantissa bits (exact in fp32).
All arithmetic is int64 masked to 32 bits, hence identical everywhere.
"""
import math
import zlib

import torch

_M32 = 0xFFFFFFFF


def _fmix32(h: torch.Tensor) -> torch.Tensor:
    h = h & _M32
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & _M32
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & _M32
    h = h ^ (h >> 16)
    return h


def name_id(name: str) -> int:
    """Stable 32-bit id of a tensor name (crc32)."""
    return zlib.crc32(name.encode()) & _M32


def uniform(shape, seed: int, tid: int, chunk: int = 1 << 24) -> torch.Tensor:
    """float32 tensor of `shape`, i.i.d.-looking uniform in [-1, 1)."""
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(n, dtype=torch.float32)
    base = ((tid * 0x85EBCA77) + (seed * 0xC2B2AE3D)) & _M32
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        i = torch.arange(lo, hi, dtype=torch.int64)
        h = _fmix32(((i * 0x9E3779B1) & _M32) + base)
        h = _fmix32(h + 0x27D4EB2F)
        # top 24 bits -> [0, 2^24) -> [-1, 1)
        out[lo:hi] = ((h >> 8).to(torch.float32) - 8388608.0) * (1.0 / 8388608.0)
    return out.view(*shape) if len(shape) else out.view(())


def randint(shape, seed: int, tid: int, low: int, high: int) -> torch.Tensor:
    """int64 tensor in [low, high)."""
    u = uniform(shape, seed, tid)
    k = ((u + 1.0) * 0.5 * (high - low)).floor().to(torch.int64) + low
    return k.clamp_(low, high - 1)


# ---------------------------------------------------------------------------
# U-Net weights
# ---------------------------------------------------------------------------
_SQRT3 = math.sqrt(3.0)


def fill_unet_state(shapes: dict, seed: int, mode: str = "stress") -> dict:
    """Generate a full state_dict for the reference U-Net key/shape map.

    shapes: {key: tuple shape} in state_dict order (479 entries for the full net).
    mode "stress": every branch contributes O(1) —
        conv/linear weights U(+-sqrt(3/fan_in)), biases U(+-0.1),
        GroupNorm gamma 1+U(+-0.3), beta U(+-0.2).
    mode "default": the reference's init *scales* (unet.py:405-426) — conv
        kaiming fan_out (std sqrt(2/fan_out)), Linear / final conv xavier
        gain 0.02, biases 0, GN 1/0, MHA in_proj xavier gain 1 — drawn from the
        hash generator instead of torch's RNG.
    The buffer time_embed.emb_coeff keeps its defining formula (unet.py:23-25).
    """
    sd = {}
    for key, shape in shapes.items():
        tid = name_id(key)
        if key.endswith("emb_coeff"):
            half = shape[0]
            sd[key] = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
            continue
        u = uniform(shape, seed, tid)
        is_norm = (".norm" in key) or key.startswith("final_conv.0.")
        if mode == "stress":
            if key.endswith("weight") and len(shape) >= 2:
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
                sd[key] = u * (_SQRT3 / math.sqrt(fan_in))
            elif is_norm and key.endswith("weight"):
                sd[key] = 1.0 + 0.3 * u
            elif is_norm and key.endswith("bias"):
                sd[key] = 0.2 * u
            else:  # biases
                sd[key] = 0.1 * u
        elif mode == "default":
            if len(shape) == 4:
                if key.startswith("final_conv.2."):
                    fan_in = shape[1] * shape[2] * shape[3]
                    fan_out = shape[0] * shape[2] * shape[3]
                    a = 0.02 * math.sqrt(6.0 / (fan_in + fan_out))
                    sd[key] = u * a
                else:
                    fan_out = shape[0] * shape[2] * shape[3]
                    sd[key] = u * (_SQRT3 * math.sqrt(2.0 / fan_out))
            elif len(shape) == 2:
                gain = 1.0 if key.endswith("in_proj_weight") else 0.02
                a = gain * math.sqrt(6.0 / (shape[0] + shape[1]))
                sd[key] = u * a
            elif is_norm and key.endswith("weight"):
                sd[key] = torch.ones(shape)
            else:
                sd[key] = torch.zeros(shape)
        else:
            raise ValueError(mode)
    return sd


def unet_inputs(batch: int, seed: int, seq: int = 32, text_dim: int = 256,
                latent_dim: int = 8, hw: int = 27, t=None):
    """Synthetic (latent/noisy latent, timesteps, text_emb) of the reference's
    I/O shapes (tests/test_dimensions.py:40-44): [B,8,27,27] f32, [B] int64,
    [B,32,256] f32.  Values ~U(+-sqrt3) (unit variance)."""
    x = uniform((batch, latent_dim, hw, hw), seed, name_id("input.x")) * _SQRT3
    text = uniform((batch, seq, text_dim), seed, name_id("input.text")) * _SQRT3
    if t is None:
        tt = randint((batch,), seed, name_id("input.t"), 0, 1000)
    else:
        tt = torch.tensor(list(t), dtype=torch.int64)
    return x, tt, text

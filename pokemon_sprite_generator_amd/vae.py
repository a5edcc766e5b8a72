"""Frozen VAE encoder / decoder on the MI355X kernels (reference: src/models/vae_decoder.py).

The VAE sits either side of the U-Net path: stage 2 encodes every image batch with the FROZEN encoder before the train
step (improved_diffusion_trainer.py:198-208,357-358) and decodes sampled latents for monitoring (:598); stage 3 and the
demo decode with it.  It is inference-only here (`torch.no_grad()` inside; parameters are containers with the
reference's names and shapes, so `load_state_dict` takes a stage-1 checkpoint's 'vae_state_dict' halves unchanged).

Same classes / constructor and forward signatures as the reference: `ResNetBlock`, `CrossAttentionBlock`, `VAEEncoder`,
`VAEDecoder`, `PokemonVAE`.  Everything runs channels-last in `compute_dtype` on the kernels of the U-Net path:
3x3 / 1x1 convolutions and the encoder's 4x4 stride-2 convolutions (psg_conv_fwd, fused bias / ReLU / Tanh / residual),
GroupNorm+SiLU, bilinear upsampling and the attention core.  Reference quirks kept, not fixed: the decoder's
cross-attention reshapes the [B, S, C] key / value projections straight to [B, heads, head_dim, S]
(vae_decoder.py:56-57 - a reinterpretation of memory, not a transpose), and the encoder returns a SAMPLED latent.
"""
import math

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import ACT_NONE, ACT_RELU, ACT_SILU, ACT_TANH, check, ptr, stream_ptr


def _prepared(cache, key, param_list, build):
    """Prepared (kernel-layout) weights of a frozen module, rebuilt when a parameter changes (version counters)."""
    stamp = tuple((p.data_ptr(), p._version) for p in param_list)
    ent = cache.get(key)
    if ent is None or ent[0] != stamp:
        ent = (stamp, build())
        cache[key] = ent
    return ent[1]


def _conv(x, wf, bias, Cin, Cout, ks, stride, pad, act=ACT_NONE, residual=None, out_dtype=None):
    """x [B,H,W,Cin] channels-last (rows 16-byte aligned) -> [B,Ho,Wo,Cout]; wf = prepared weight [Cout][Kpad]."""
    lib = ops._lib_for(x)
    xr, ldx = ops._rows(x)
    B, Hi, Wi = x.shape[0], x.shape[1], x.shape[2]
    Ho, Wo = (Hi + 2 * pad - ks) // stride + 1, (Wi + 2 * pad - ks) // stride + 1
    y = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    res_r, ld_res = (None, 0) if residual is None else ops._rows(residual)
    ops._conv_launch(lib, x.dtype, xr, ldx, wf, 0, y, Cout, (B, Hi, Wi, Ho, Wo, ks, stride, pad), Cin, Cout, bias=bias,
                     residual=res_r, ld_res=ld_res, act=act)
    return y


def _prep(w, dtype, pad_in=0, pad_out=0):
    """fp32 OIHW parameter -> prepared forward weight [O'][Kpad] in `dtype`; optional zero padding of Cin / Cout."""
    lib = ops._lib_for(w)
    O, I, kh, kw = w.shape
    src = w.detach().float()
    if pad_in or pad_out:
        src = torch.nn.functional.pad(src, (0, 0, 0, 0, 0, pad_in, 0, pad_out))
        O, I = O + pad_out, I + pad_in
    src = src.contiguous(memory_format=torch.channels_last)           # OHWI memory: the layout every kernel-side path takes
    code = _lib.dtype_code(dtype)
    kp = lib.psg_kpad(kh * kw * I, code)
    wf = torch.empty((O, kp), dtype=dtype, device=w.device)
    check(lib.psg_prep_weight(ptr(src), _lib.dtype_code(torch.float32), ops.W_OHWI, ptr(wf), None, O, I, kh, code, stream_ptr()),
          "psg_prep_weight")
    return wf


class ResNetBlock(nn.Module):
    """vae_decoder.py:8-32."""

    def __init__(self, in_channels: int, out_channels: int, groups: int = 32, dropout: float = 0.0):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.norm2 = nn.GroupNorm(groups, out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        self.dropout = nn.Dropout(dropout)
        self.shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1) if in_channels != out_channels else nn.Identity()
        self._cache = {}

    def nhwc(self, x):
        dt = x.dtype
        cin, cout = self.conv1.in_channels, self.conv1.out_channels
        w1 = _prepared(self._cache, ("c1", dt), [self.conv1.weight], lambda: _prep(self.conv1.weight, dt))
        w2 = _prepared(self._cache, ("c2", dt), [self.conv2.weight], lambda: _prep(self.conv2.weight, dt))
        h = ops.group_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.num_groups, self.norm1.eps, silu=True)
        h = _conv(h, w1, self.conv1.bias, cin, cout, 3, 1, 1)
        h = ops.group_norm(h, self.norm2.weight, self.norm2.bias, self.norm2.num_groups, self.norm2.eps, silu=True)
        if isinstance(self.shortcut, nn.Conv2d):
            ws = _prepared(self._cache, ("sc", dt), [self.shortcut.weight], lambda: _prep(self.shortcut.weight, dt))
            skip = _conv(x, ws, self.shortcut.bias, cin, cout, 1, 1, 0)
        else:
            skip = x
        return _conv(h, w2, self.conv2.bias, cout, cout, 3, 1, 1, residual=skip)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        dt = getattr(self, "compute_dtype", torch.float32)
        return ops.nhwc_to_nchw(self.nhwc(ops.nchw_to_nhwc(x, dt)))


class CrossAttentionBlock(nn.Module):
    """vae_decoder.py:33-65 - the decoder's hand-rolled text cross-attention."""

    def __init__(self, channels: int, text_dim: int, num_heads: int = 8):
        super().__init__()
        self.channels, self.text_dim, self.num_heads = channels, text_dim, num_heads
        self.head_dim = channels // num_heads
        self.norm = nn.GroupNorm(32, channels)
        self.q = nn.Conv2d(channels, channels, kernel_size=1)
        self.k = nn.Linear(text_dim, channels)
        self.v = nn.Linear(text_dim, channels)
        self.proj = nn.Conv2d(channels, channels, kernel_size=1)
        self._cache = {}

    def nhwc(self, x, text):
        """x [B,H,W,C], text [B,S,text_dim] in the compute dtype."""
        dt = x.dtype
        B, H, W, C = x.shape
        S = text.shape[1]
        wq = _prepared(self._cache, ("q", dt), [self.q.weight], lambda: _prep(self.q.weight, dt))
        wp = _prepared(self._cache, ("p", dt), [self.proj.weight], lambda: _prep(self.proj.weight, dt))
        wk = _prepared(self._cache, ("k", dt), [self.k.weight], lambda: _prep(self.k.weight[:, :, None, None], dt))
        wv = _prepared(self._cache, ("v", dt), [self.v.weight], lambda: _prep(self.v.weight[:, :, None, None], dt))
        xn = ops.group_norm(x, self.norm.weight, self.norm.bias, self.norm.num_groups, self.norm.eps)
        q = _conv(xn, wq, self.q.bias, C, C, 1, 1, 0).reshape(B, H * W, C)
        t4 = text.reshape(B, S, 1, self.text_dim)
        k = _conv(t4, wk, self.k.bias, self.text_dim, C, 1, 1, 0).reshape(B, S, C)
        v = _conv(t4, wv, self.v.bias, self.text_dim, C, 1, 1, 0).reshape(B, S, C)
        # vae_decoder.py:56-57: `.reshape(b, heads, head_dim, -1)` of a [b, S, C] tensor reads its memory as [C][S]:
        # key token s' of channel c is element c*S + s' of the projection's buffer.  The attention kernel wants [S][C] rows.
        kv = torch.cat([k.reshape(B, C, S).transpose(1, 2), v.reshape(B, C, S).transpose(1, 2)], dim=-1).contiguous()
        o = ops.attention_cross(q, kv, self.num_heads)             # softmax(q^T k / sqrt(head_dim)) v, per head
        y = _conv(o.reshape(B, H, W, C), wp, self.proj.bias, C, C, 1, 1, 0, residual=x)
        return y

    @torch.no_grad()
    def forward(self, x: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
        dt = getattr(self, "compute_dtype", torch.float32)
        return ops.nhwc_to_nchw(self.nhwc(ops.nchw_to_nhwc(x, dt), text_emb.to(dt)))


class VAEEncoder(nn.Module):
    """vae_decoder.py:68-125: 215 -> 107 -> 53 -> 27 by three 4x4 stride-2 convolutions (+ReLU, ResNetBlock), four more
    ResNetBlocks up to 512 channels, mu / logvar 3x3 projections, reparameterised sample."""

    def __init__(self, input_channels: int = 3, latent_dim: int = 8, compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        self.latent_dim = latent_dim
        self.encoder = nn.Sequential(
            nn.Conv2d(input_channels, 32, kernel_size=4, stride=2, padding=1), nn.ReLU(), ResNetBlock(32, 32),
            nn.Conv2d(32, 64, kernel_size=4, stride=2, padding=1), nn.ReLU(), ResNetBlock(64, 64),
            nn.Conv2d(64, 128, kernel_size=4, stride=2, padding=2), nn.ReLU(), ResNetBlock(128, 128),
            ResNetBlock(128, 256), ResNetBlock(256, 256), ResNetBlock(256, 512), ResNetBlock(512, 512),
        )
        self.mu_proj = nn.Conv2d(512, latent_dim, kernel_size=3, padding=1)
        self.logvar_proj = nn.Conv2d(512, latent_dim, kernel_size=3, padding=1)
        self.compute_dtype = compute_dtype
        self._cache = {}

    @torch.no_grad()
    def forward(self, x: torch.Tensor, eps: torch.Tensor = None, generator: torch.Generator = None):
        """x [B,3,H,W] -> (latent, mu, logvar), each [B,latent_dim,h,w] fp32 (215 -> 27).  `eps` (default: randn of mu's
        shape, from `generator` when given - a data-parallel rank's own stream) is the reparameterisation noise the
        reference draws with randn_like (:121)."""
        if not x.is_cuda:
            raise _lib.PsgError("VAEEncoder (MI355X build) needs GPU tensors; there is no CPU fallback")
        dt = self.compute_dtype
        lib = ops._lib_for(x)
        B, Cin = x.shape[0], x.shape[1]
        # image -> channels-last with the channel count padded to one 16-byte chunk (zeros; the weights are padded alike)
        cpad = (-Cin) % 8
        h = torch.zeros((B, x.shape[2], x.shape[3], Cin + cpad), dtype=dt, device=x.device)
        xc = x.detach().contiguous().float()
        check(lib.psg_nchw_to_nhwc(ptr(xc), ptr(h), Cin + cpad, B, Cin, x.shape[2] * x.shape[3], _lib.dtype_code(dt), stream_ptr()),
              "psg_nchw_to_nhwc")
        for i, m in enumerate(self.encoder):
            if isinstance(m, nn.Conv2d):
                pin = cpad if i == 0 else 0
                wf = _prepared(self._cache, (i, dt), [m.weight], lambda m=m, pin=pin: _prep(m.weight, dt, pad_in=pin))
                h = _conv(h, wf, m.bias, m.in_channels + pin, m.out_channels, 4, 2, m.padding[0], act=ACT_RELU)   # Conv2d + the ReLU after it
            elif isinstance(m, ResNetBlock):
                h = m.nhwc(h)
        wm = _prepared(self._cache, ("mu", dt), [self.mu_proj.weight], lambda: _prep(self.mu_proj.weight, dt))
        wl = _prepared(self._cache, ("lv", dt), [self.logvar_proj.weight], lambda: _prep(self.logvar_proj.weight, dt))
        mu = ops.nhwc_to_nchw(_conv(h, wm, self.mu_proj.bias, 512, self.latent_dim, 3, 1, 1))
        logvar = ops.nhwc_to_nchw(_conv(h, wl, self.logvar_proj.bias, 512, self.latent_dim, 3, 1, 1))
        if eps is None:
            eps = torch.randn_like(mu) if generator is None else torch.randn(mu.shape, dtype=mu.dtype, device=mu.device, generator=generator)
        eps = eps.to(device=mu.device, dtype=torch.float32).contiguous()
        latent = torch.empty_like(mu)
        check(lib.psg_reparam_f32(ptr(mu), ptr(logvar), ptr(eps), ptr(latent), mu.numel(), stream_ptr()), "psg_reparam_f32")
        return latent, mu, logvar


class VAEDecoder(nn.Module):
    """vae_decoder.py:128-222: 27 -> 54 -> 108 -> 215 with a text cross-attention in every block."""

    def __init__(self, latent_dim: int = 8, text_dim: int = 256, output_channels: int = 3, compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        self.latent_dim, self.text_dim = latent_dim, text_dim
        self.latent_proj = nn.Conv2d(latent_dim, 512, kernel_size=3, padding=1)
        chans = [(512, 512), (512, 256), (256, 128), (128, 64), (64, 32)]
        for i, (cin, cout) in enumerate(chans, start=1):
            setattr(self, f"block{i}_resnet1", ResNetBlock(cin, cout))
            setattr(self, f"block{i}_attn", CrossAttentionBlock(cout, text_dim))
            setattr(self, f"block{i}_resnet2", ResNetBlock(cout, cout))
            if i in (2, 3):
                setattr(self, f"block{i}_upsample", nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
            elif i == 4:
                setattr(self, f"block{i}_upsample", nn.Upsample(size=(215, 215), mode="bilinear", align_corners=False))
        self.final_conv = nn.Sequential(nn.GroupNorm(8, 32), nn.SiLU(), nn.Conv2d(32, output_channels, kernel_size=3, padding=1), nn.Tanh())
        self.compute_dtype = compute_dtype
        self._cache = {}

    @torch.no_grad()
    def forward(self, latent: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
        if not latent.is_cuda:
            raise _lib.PsgError("VAEDecoder (MI355X build) needs GPU tensors; there is no CPU fallback")
        dt = self.compute_dtype
        text = text_emb.detach().to(device=latent.device, dtype=dt).contiguous()
        wl = _prepared(self._cache, ("lp", dt), [self.latent_proj.weight], lambda: _prep(self.latent_proj.weight, dt))
        x = _conv(ops.nchw_to_nhwc(latent, dt), wl, self.latent_proj.bias, self.latent_dim, 512, 3, 1, 1)
        for i in range(1, 6):
            x = getattr(self, f"block{i}_resnet1").nhwc(x)
            x = getattr(self, f"block{i}_attn").nhwc(x, text)
            x = getattr(self, f"block{i}_resnet2").nhwc(x)
            up = getattr(self, f"block{i}_upsample", None)
            if up is not None:
                size = up.size if up.size is not None else (int(x.shape[1] * up.scale_factor), int(x.shape[2] * up.scale_factor))
                x = ops.upsample_bilinear(x, size)
        gn, conv = self.final_conv[0], self.final_conv[2]
        x = ops.group_norm(x, gn.weight, gn.bias, gn.num_groups, gn.eps, silu=True)
        cout = conv.out_channels
        opad = (-cout) % 4                                           # 3 image channels -> 4 output columns (zero rows of weight)
        wf = _prepared(self._cache, ("fc", dt), [conv.weight, conv.bias],
                       lambda: (_prep(conv.weight, dt, pad_out=opad), torch.nn.functional.pad(conv.bias.detach().float(), (0, opad)).contiguous()))
        y = _conv(x, wf[0], wf[1], conv.in_channels, cout + opad, 3, 1, 1, act=ACT_TANH)
        lib = ops._lib_for(y)
        B, H, W, _ = y.shape
        img = torch.empty((B, cout, H, W), dtype=torch.float32, device=y.device)
        check(lib.psg_nhwc_to_nchw(ptr(y), cout + opad, ptr(img), B, cout, H * W, _lib.dtype_code(dt), stream_ptr()), "psg_nhwc_to_nchw")
        return img


class PokemonVAE(nn.Module):
    """vae_decoder.py:225-291."""

    def __init__(self, latent_dim: int = 8, text_dim: int = 256, compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        self.latent_dim, self.text_dim = latent_dim, text_dim
        self.encoder = VAEEncoder(input_channels=3, latent_dim=latent_dim, compute_dtype=compute_dtype)
        self.decoder = VAEDecoder(latent_dim=latent_dim, text_dim=text_dim, output_channels=3, compute_dtype=compute_dtype)

    @torch.no_grad()
    def forward(self, images: torch.Tensor, text_emb: torch.Tensor, mode: str = "train") -> dict:
        if mode == "sample" or images is None:
            latent = torch.randn(text_emb.size(0), self.latent_dim, 27, 27, device=text_emb.device)
            mu = logvar = None
        else:
            latent, mu, logvar = self.encoder(images)
            if mode == "generate":
                latent = mu
        return {"reconstructed": self.decoder(latent, text_emb), "latent": latent, "mu": mu, "logvar": logvar}

    def encode(self, images):
        return self.encoder(images)

    def decode(self, latent, text_emb):
        return self.decoder(latent, text_emb)

    def sample(self, batch_size: int, text_emb: torch.Tensor, device: torch.device) -> torch.Tensor:
        return self.decoder(torch.randn(batch_size, self.latent_dim, 27, 27, device=device), text_emb)

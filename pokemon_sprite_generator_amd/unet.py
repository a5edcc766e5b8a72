"""Host-side mirror of the reference U-Net (src/models/unet.py) on the MI355X kernels.

Same class names, constructor signatures, forward signatures and state_dict
(479 entries, identical keys/shapes, NCHW/OIHW fp32) as the reference, so a
checkpoint's 'unet_state_dict' loads either way and `UNet` substitutes at the
reference's import sites.  torch.nn modules are used ONLY as parameter
containers (which also reproduces the reference's initialisation RNG stream);
every forward/backward FLOP runs in libpsg_hip.so through `ops`.

Internally activations are channels-last ([B,H,W,C] == token layout [B,L,C]) in
`compute_dtype` (fp32: strict parity path; bf16: MFMA throughput path); GroupNorm
statistics, softmax, accumulators, parameters and their gradients stay fp32.
"""
import math

import os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_GELU, ACT_NONE, ACT_SILU

_FUSED_FFN = os.environ.get("PSG_FFN_FUSE", "1") != "0"
_PROJ_GROUP = os.environ.get("PSG_PROJ_GROUP", "1") != "0"         # 0: every ResBlock runs its own time_proj / text_proj GEMMs (A/B)
_CONCAT_SLOTS = os.environ.get("PSG_CONCAT_SLOTS", "1") != "0"     # 0: torch.cat in the decoder, autograd sums the skip gradients (A/B)

ATTN_DROPOUT = 0.05   # nn.MultiheadAttention(dropout=0.05), nn.Dropout(0.05): unet.py:160-187


def _norm_groups(channels: int) -> int:
    """Largest group count <= 32 dividing `channels` (unet.py:70-76,151-153)."""
    g = min(32, channels)
    while channels % g and g > 1:
        g -= 1
    return max(1, g)


class _SeedStream:
    """Per-call dropout seeds: (torch seed, data-parallel rank, running counter) -> 64-bit seed per dropout site.
    The rank is mixed in so that ranks seeded alike (same initial weights) still draw different masks for their
    different batch shards; `rank` is read from the default process group unless set explicitly."""
    counter = 0
    rank = None

    @classmethod
    def _rank(cls):
        if cls.rank is not None:
            return int(cls.rank)
        d = torch.distributed
        return d.get_rank() if (d.is_available() and d.is_initialized()) else 0

    @classmethod
    def next(cls):
        cls.counter += 1
        base = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        return (base * 0x9E3779B97F4A7C15 + cls._rank() * 0xBF58476D1CE4E5B9 + cls.counter * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


def _compute_dtype_of(module, default=torch.float32):
    return getattr(module, "compute_dtype", default)


class TimestepEmbedding(nn.Module):
    """unet.py:12-55 — sinusoidal features + 3-layer SiLU MLP."""

    def __init__(self, embedding_dim: int = 128, max_time: int = 1000):
        super().__init__()
        self.embedding_dim, self.max_time = embedding_dim, max_time
        half = embedding_dim // 2
        self.register_buffer("emb_coeff", torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))))
        widths = [embedding_dim, embedding_dim * 4, embedding_dim * 4, embedding_dim]
        layers = []
        for i in range(3):
            layers.append(nn.Linear(widths[i], widths[i + 1]))
            if i < 2:
                layers.append(nn.SiLU())
        self.time_mlp = nn.Sequential(*layers)
        self.compute_dtype = torch.float32

    def embed(self, timesteps, dtype):
        e = ops.timestep_sinusoid(timesteps, self.emb_coeff, dtype)
        m = self.time_mlp
        h = ops.linear(e, m[0].weight, m[0].bias, act=ACT_SILU)
        h = ops.linear(h, m[2].weight, m[2].bias, act=ACT_SILU)
        return ops.linear(h, m[4].weight, m[4].bias)

    def forward(self, timesteps: torch.Tensor) -> torch.Tensor:
        return self.embed(timesteps, self.compute_dtype).float()


class ResBlock(nn.Module):
    """unet.py:58-132 — GN+SiLU, 3x3 conv (+time/text per-sample add), GN+SiLU, 3x3 conv, residual."""

    def __init__(self, in_channels: int, out_channels: int, time_emb_dim: int = 128, text_emb_dim: int = 256,
                 dropout: float = 0.0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = nn.GroupNorm(_norm_groups(in_channels), in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.time_proj = nn.Linear(time_emb_dim, out_channels)
        self.text_proj = nn.Linear(text_emb_dim, out_channels)
        self.norm2 = nn.GroupNorm(_norm_groups(out_channels), out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        self.dropout = nn.Dropout(dropout)      # p = 0.0 in every reference instance: identity
        self.skip_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1) if in_channels != out_channels else nn.Identity()
        self.compute_dtype = torch.float32

    def nhwc(self, x, temb, pooled, out=None, ra=None):
        """x [B,H,W,Cin], temb [B,128], pooled [B,256] in the compute dtype; `out`: ops.OutSlot for the result; `ra`: the
        per-sample additive vector when the caller computed it for all blocks at once (ops.ProjGroup)."""
        # (h, x): x's second consumer below is the skip path - its gradient joins inside the GroupNorm backward
        h, x = ops.group_norm_split(x, self.norm1.weight, self.norm1.bias, self.norm1.num_groups, self.norm1.eps, silu=True)
        # per-sample additive vector: time_proj(temb) + text_proj(pooled)  (unet.py:119-124), fused into conv1's epilogue
        if ra is None:
            ra = ops.linear(temb, self.time_proj.weight, self.time_proj.bias)
            ra = ops.linear(pooled, self.text_proj.weight, self.text_proj.bias, residual=ra)
        h = ops.conv2d(h, self.conv1.weight, self.conv1.bias, rowadd=ra)
        h = ops.group_norm(h, self.norm2.weight, self.norm2.bias, self.norm2.num_groups, self.norm2.eps, silu=True)
        if isinstance(self.skip_conv, nn.Conv2d):
            skip = ops.conv2d(x, self.skip_conv.weight, self.skip_conv.bias)
        else:
            skip = x
        return ops.conv2d(h, self.conv2.weight, self.conv2.bias, residual=skip, out=out)

    def forward(self, x: torch.Tensor, time_emb: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
        dt = self.compute_dtype
        y = self.nhwc(_to_cl(x, dt), _to_dt(time_emb, dt), _to_dt(text_emb, dt))
        return ops.nhwc_to_nchw(y)


class CrossAttentionBlock(nn.Module):
    """unet.py:135-260 — x += 0.7*SelfAttn(GN(x)); x += 0.8*CrossAttn(GN(x), text_proj(text)); x += 0.6*FFN(x)."""

    def __init__(self, channels: int, text_dim: int, num_heads: int = 8):
        super().__init__()
        assert channels % num_heads == 0, f"channels ({channels}) must be divisible by num_heads ({num_heads})"
        self.channels, self.text_dim, self.num_heads = channels, text_dim, num_heads
        self.head_dim = channels // num_heads
        g = _norm_groups(channels)
        self.norm1 = nn.GroupNorm(g, channels, eps=1e-6)
        self.norm2 = nn.GroupNorm(g, channels, eps=1e-6)
        self.self_attn = nn.MultiheadAttention(embed_dim=channels, num_heads=num_heads, dropout=ATTN_DROPOUT, batch_first=True)
        self.cross_attn = nn.MultiheadAttention(embed_dim=channels, num_heads=num_heads, dropout=ATTN_DROPOUT, batch_first=True)
        self.text_proj = nn.Linear(text_dim, channels)
        _small_xavier(self.text_proj)
        self.ffn = nn.Sequential(nn.Linear(channels, channels * 2), nn.GELU(), nn.Dropout(ATTN_DROPOUT),
                                 nn.Linear(channels * 2, channels), nn.Dropout(ATTN_DROPOUT))
        for layer in self.ffn:
            if isinstance(layer, nn.Linear):
                _small_xavier(layer)
        self.compute_dtype = torch.float32

    def nhwc(self, x, text, out=None):
        """x [B,H,W,C] (== tokens [B,L,C]); text [B,S,text_dim]; compute dtype; `out`: ops.OutSlot for the result."""
        B, H, W, C = x.shape
        if out is not None:
            out = ops.OutSlot(out.view.reshape(B, H * W, C))        # (a view: the slot's H and W are adjacent in memory)
        p = ATTN_DROPOUT if self.training else 0.0
        seeds = [_SeedStream.next() for _ in range(4)] if p > 0 else [0, 0, 0, 0]
        tok = x.reshape(B, H * W, C)
        sa, ca = self.self_attn, self.cross_attn
        # self-attention (unet.py:212-221)
        xn, tok = ops.group_norm_split(tok, self.norm1.weight, self.norm1.bias, self.norm1.num_groups, self.norm1.eps)
        qkv = ops.linear(xn, sa.in_proj_weight, sa.in_proj_bias)
        o = ops.attention_self(qkv, self.num_heads, p, seeds[0])
        tok = ops.linear(o, sa.out_proj.weight, sa.out_proj.bias, residual=tok, alpha=0.7)
        # cross-attention (unet.py:229-239)
        xn, tok = ops.group_norm_split(tok, self.norm2.weight, self.norm2.bias, self.norm2.num_groups, self.norm2.eps)
        tp = ops.linear(text, self.text_proj.weight, self.text_proj.bias)
        q, kv = ops.cross_in_proj(xn, tp, ca.in_proj_weight, ca.in_proj_bias)
        o = ops.attention_cross(q, kv, self.num_heads, p, seeds[1])
        tok = ops.linear(o, ca.out_proj.weight, ca.out_proj.bias, residual=tok, alpha=0.8)
        # feed-forward, no norm (unet.py:247-251)
        if _FUSED_FFN:
            tok = ops.ffn(tok, self.ffn[0].weight, self.ffn[0].bias, self.ffn[3].weight, self.ffn[3].bias, 0.6, p, seeds[2], seeds[3], out=out)
        else:                                          # two generic nodes (PSG_FFN_FUSE=0: A/B and debugging)
            h = ops.linear(tok, self.ffn[0].weight, self.ffn[0].bias, act=ACT_GELU, drop_p=p, seed=seeds[2])
            tok = ops.linear(h, self.ffn[3].weight, self.ffn[3].bias, residual=tok, alpha=0.6, drop_p=p, seed=seeds[3], out=out)
        return tok.reshape(B, H, W, C)

    def forward(self, x: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
        dt = self.compute_dtype
        return ops.nhwc_to_nchw(self.nhwc(_to_cl(x, dt), _to_dt(text_emb, dt)))


class UNetBlock(nn.Module):
    """unet.py:263-301."""

    def __init__(self, in_channels: int, out_channels: int, time_emb_dim: int = 128, text_emb_dim: int = 256,
                 has_attention: bool = True, num_heads: int = 8):
        super().__init__()
        self.has_attention = has_attention
        self.res_block = ResBlock(in_channels, out_channels, time_emb_dim, text_emb_dim)
        if has_attention:
            self.attn_block = CrossAttentionBlock(out_channels, text_emb_dim, num_heads)
        self.compute_dtype = torch.float32

    def nhwc(self, x, temb, pooled, text, out=None, ra=None):
        if not self.has_attention:
            return self.res_block.nhwc(x, temb, pooled, out=out, ra=ra)
        x = self.res_block.nhwc(x, temb, pooled, ra=ra)
        return self.attn_block.nhwc(x, text, out=out)

    def forward(self, x, time_emb, text_emb, text_seq):
        dt = self.compute_dtype
        y = self.nhwc(_to_cl(x, dt), _to_dt(time_emb, dt), _to_dt(text_emb, dt), _to_dt(text_seq, dt))
        return ops.nhwc_to_nchw(y)


# level: (channels, spatial, attention)
_LEVELS = ((320, 27, False), (640, 14, True), (1280, 7, True), (1280, 4, True))


class UNet(nn.Module):
    """unet.py:304-509 — [B,8,27,27] noisy latent, [B] timesteps, [B,S,text_dim] text -> predicted noise.

    `compute_dtype` (extra keyword, default fp32) selects the arithmetic of the
    matrix kernels: torch.float32 (exact-fp32 MFMA, parity path) or
    torch.bfloat16 (bf16 MFMA with fp32 accumulation).
    """

    def __init__(self, latent_dim: int = 8, text_dim: int = 256, time_emb_dim: int = 128, num_heads: int = 8,
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        self.latent_dim, self.text_dim, self.time_emb_dim = latent_dim, text_dim, time_emb_dim
        mk = lambda cin, cout, attn: UNetBlock(cin, cout, time_emb_dim, text_dim, has_attention=attn, num_heads=num_heads)
        self.time_embed = TimestepEmbedding(time_emb_dim)
        self.text_pool = nn.AdaptiveAvgPool1d(1)
        self.init_conv = nn.Conv2d(latent_dim, _LEVELS[0][0], kernel_size=3, padding=1)
        prev = _LEVELS[0][0]
        for lvl, (ch, _, attn) in enumerate(_LEVELS):          # encoder (creation order == reference: RNG parity)
            if lvl > 0:
                setattr(self, f"downsample{lvl}", nn.Conv2d(prev, ch, kernel_size=3, stride=2, padding=1))
            setattr(self, f"enc_block{lvl}", nn.ModuleList([mk(ch, ch, attn), mk(ch, ch, attn)]))
            prev = ch
        self.middle_block = mk(prev, prev, True)
        for lvl in (3, 2, 1, 0):                               # decoder
            ch, _, attn = _LEVELS[lvl]
            setattr(self, f"dec_block{lvl}", nn.ModuleList([mk(2 * ch, ch, attn), mk(2 * ch, ch, attn)]))
            if lvl > 0:
                nxt, size, _ = _LEVELS[lvl - 1]
                setattr(self, f"upsample{lvl}", nn.Sequential(nn.Upsample(size=(size, size), mode="bilinear", align_corners=False),
                                                              nn.Conv2d(ch, nxt, kernel_size=3, padding=1)))
        self.final_conv = nn.Sequential(nn.GroupNorm(32, _LEVELS[0][0]), nn.SiLU(),
                                        nn.Conv2d(_LEVELS[0][0], latent_dim, kernel_size=3, padding=1))
        self._initialize_weights()
        self.set_compute_dtype(compute_dtype)

    def blocks_in_order(self):
        """The 17 UNetBlocks in forward order."""
        out = []
        for lvl in range(4):
            out += list(getattr(self, f"enc_block{lvl}"))
        out.append(self.middle_block)
        for lvl in (3, 2, 1, 0):
            out += list(getattr(self, f"dec_block{lvl}"))
        return out

    def arena_layout(self, params):
        """Memory order of the flat parameter / gradient arenas (optim._arena_offsets): the ResBlocks' time_proj weights next
        to each other in block order, then their biases, text_proj likewise (ops.ProjGroup: two GEMMs instead of 34), then
        everything else in parameter order.  They come FIRST: their gradients are the last backward produces, like those of the
        time-embedding MLP and init_conv that follow - the data-parallel buckets leave in reverse memory order."""
        params = list(params)
        pidx = {id(p): i for i, p in enumerate(params)}
        head = []
        for kind, attr in (("time_proj", "weight"), ("time_proj", "bias"), ("text_proj", "weight"), ("text_proj", "bias")):
            for b in self.blocks_in_order():
                t = getattr(getattr(b.res_block, kind), attr)
                if id(t) not in pidx:
                    return None
                head.append(pidx[id(t)])
        taken = set(head)
        return head + [i for i in range(len(params)) if i not in taken]

    def bind_proj_group(self, param_arena, grad_arena):
        """Called by the stepper once both arenas exist; `None` arenas unbind."""
        if getattr(self, "_proj_group", None) is not None:
            self._proj_group.release()
        self._proj_group = None
        if param_arena is not None and _PROJ_GROUP:
            g = ops.ProjGroup([b.res_block for b in self.blocks_in_order()], param_arena, grad_arena)
            self._proj_group = g if g.ok else None

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"compute_dtype must be torch.float32 or torch.bfloat16, got {dtype}")
        for m in self.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = dtype
        self.compute_dtype = dtype
        return self

    def _initialize_weights(self):
        """Same initial distribution (and RNG stream) as unet.py:405-426."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                _small_xavier(m)
            elif isinstance(m, nn.GroupNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        for m in self.final_conv.modules():
            if isinstance(m, nn.Conv2d):
                _small_xavier(m)

    def forward(self, noisy_latent: torch.Tensor, timesteps: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
        dt = self.compute_dtype
        if noisy_latent.dim() != 4 or noisy_latent.shape[1] != self.latent_dim:
            raise ValueError(f"noisy_latent must be [B,{self.latent_dim},H,W], got {tuple(noisy_latent.shape)}")
        if text_emb.dim() != 3 or text_emb.shape[-1] != self.text_dim or text_emb.shape[0] != noisy_latent.shape[0]:
            raise ValueError(f"text_emb must be [B,S,{self.text_dim}], got {tuple(text_emb.shape)}")
        x = ops.nchw_to_nhwc(noisy_latent, dt)
        temb = self.time_embed.embed(timesteps, dt)
        pooled, text = ops.text_pool(text_emb, dt)
        x = ops.conv2d(x, self.init_conv.weight, self.init_conv.bias)
        # per-sample additive vectors of all 17 ResBlocks in two GEMMs when the parameters live in a stepper's arenas
        grp = getattr(self, "_proj_group", None)
        ras = iter(grp.rowadds(temb, pooled)) if (grp is not None and grp.usable()) else None
        nra = (lambda: next(ras)) if ras is not None else (lambda: None)
        skips = []
        slots = _CONCAT_SLOTS and all(len(getattr(self, f"dec_block{lvl}")) == 2 for lvl in range(4))
        for lvl in range(4):
            if lvl > 0:
                ds = getattr(self, f"downsample{lvl}")
                x = ops.conv2d(x, ds.weight, ds.bias, stride=2)
            for blk in getattr(self, f"enc_block{lvl}"):
                x = blk.nhwc(x, temb, pooled, text, ra=nra())
            if slots:
                # the skip has three consumers (next stage, two decoder blocks): their gradients are summed in one pass
                x, s0, s1 = ops.fan3(x)
                skips.append((s0, s1))
            else:
                skips.append(x)
        if not slots:
            x = self.middle_block.nhwc(x, temb, pooled, text, ra=nra())
            for lvl in (3, 2, 1, 0):
                skip = skips.pop()
                for blk in getattr(self, f"dec_block{lvl}"):
                    x = blk.nhwc(torch.cat([x, skip], dim=-1), temb, pooled, text, ra=nra())   # same skip for both blocks (unet.py:480-504)
                if lvl > 0:
                    up = getattr(self, f"upsample{lvl}")
                    x = ops.upsample_bilinear(x, up[0].size)
                    x = ops.conv2d(x, up[1].weight, up[1].bias)
        else:
            # torch.cat([x, skip]) without the concat (unet.py:480-504; same skip for both blocks of a level): the skip half
            # is copied into a [.., C1 + C2] buffer when that buffer is made, and the op that produces x writes it straight
            # into the other half (ops.ConcatSlot / ops.OutSlot) - half the concat's traffic, no CatArrayBatchedCopy.
            slot = ops.ConcatSlot(skips[-1][0], self.middle_block.res_block.out_channels)
            x = self.middle_block.nhwc(x, temb, pooled, text, out=slot.out, ra=nra())
            for lvl in (3, 2, 1, 0):
                s0, s1 = skips.pop()
                b0, b1 = getattr(self, f"dec_block{lvl}")
                xc = slot.cat(x, s0)
                slot = ops.ConcatSlot(s1, b0.res_block.out_channels)
                x = b0.nhwc(xc, temb, pooled, text, out=slot.out, ra=nra())
                x = b1.nhwc(slot.cat(x, s1), temb, pooled, text, ra=nra())
                if lvl > 0:
                    up = getattr(self, f"upsample{lvl}")
                    x = ops.upsample_bilinear(x, up[0].size)
                    slot = ops.ConcatSlot(skips[-1][0], up[1].out_channels)
                    x = ops.conv2d(x, up[1].weight, up[1].bias, out=slot.out)
        gn, conv = self.final_conv[0], self.final_conv[2]
        x = ops.group_norm(x, gn.weight, gn.bias, gn.num_groups, gn.eps, silu=True)
        x = ops.conv2d(x, conv.weight, conv.bias)
        return ops.nhwc_to_nchw(x)


def _small_xavier(m):
    nn.init.xavier_uniform_(m.weight, gain=0.02)
    if m.bias is not None:
        nn.init.zeros_(m.bias)


def _to_dt(t, dtype):
    return t if t.dtype == dtype else t.to(dtype)


class _ToCL(torch.autograd.Function):
    """[B,C,H,W] fp32 -> [B,H,W,C] compute dtype, differentiable (standalone block API only)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.shape = tuple(x.shape)
        return ops.nchw_to_nhwc(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return ops.nhwc_to_nchw(dy.contiguous()), None


def _to_cl(x, dtype):
    return _ToCL.apply(x, dtype)


def test_unet():
    """Shape smoke test, like the reference's (unet.py:512-542)."""
    dev = torch.device("cuda")
    unet = UNet(latent_dim=8, text_dim=256).to(dev)
    with torch.no_grad():
        out = unet(torch.randn(4, 8, 27, 27, device=dev), torch.randint(0, 1000, (4,), device=dev), torch.randn(4, 32, 256, device=dev))
    assert out.shape == (4, 8, 27, 27)
    return True

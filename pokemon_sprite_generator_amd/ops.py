"""torch.autograd.Function wrappers over the C ABI (libpsg_hip.so).

All activations are channels-last: a conv tensor is [B, H, W, C], a token tensor
[B, L, C]; the last dim is contiguous and rows have a uniform stride.  Every op
here launches hand-written HIP kernels on torch's current stream; PyTorch only
owns the memory.  There is no eager fallback: tensors must be on a GPU.
"""
import ctypes as C
import os
import struct
import threading

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_SILU, ConvDesc, WgradDesc, check, dtype_code, ptr, stream_ptr

__all__ = ["conv2d", "linear", "group_norm", "group_norm_split", "attention_self", "attention_cross", "cross_in_proj", "upsample_bilinear",
           "nchw_to_nhwc", "nhwc_to_nchw", "text_pool", "timestep_sinusoid", "WeightCache", "ACT_NONE", "ACT_SILU", "ACT_GELU"]


def _lib_for(t):
    if not t.is_cuda:
        raise _lib.PsgError("pokemon_sprite_generator_amd ops need GPU tensors: the HIP kernels are the only implementation")
    return _lib.init(t.device.index if t.device.index is not None else torch.cuda.current_device())


class RowCopies:
    """Counter of the hidden `.contiguous()` copies `_rows` had to make (irregular strides / misaligned rows): each one is
    a full extra pass over an activation through HBM.  The U-Net path is expected to make none (asserted by
    tests/test_unet_gpu.py); PSG_WARN_COPIES=1 logs the shape and strides of every offender."""
    count = 0
    bytes = 0
    warn = os.environ.get("PSG_WARN_COPIES", "0") != "0"

    @classmethod
    def note(cls, t):
        cls.count += 1
        cls.bytes += t.numel() * t.element_size()
        if cls.warn:
            import warnings
            warnings.warn(f"psg ops: layout copy of a {tuple(t.shape)} tensor with strides {tuple(t.stride())} (ptr%16={t.data_ptr() % 16})")


def _rows(t):
    """Collapse [..., C] with uniform row stride to (rows, ld); copies (and counts it, RowCopies) if the layout is irregular."""
    if t.stride(-1) != 1:
        RowCopies.note(t)
        t = t.contiguous()
    C_ = t.shape[-1]
    ld = t.stride(-2) if t.dim() >= 2 else C_
    ok = True
    expect = ld
    for i in range(t.dim() - 2, -1, -1):
        if t.shape[i] != 1 and t.stride(i) != expect:
            ok = False
            break
        expect = expect * t.shape[i]
    if not ok or (t.data_ptr() % 16) != 0 or (ld % 8) != 0:
        RowCopies.note(t)
        t = t.contiguous()
        ld = C_
    return t, ld


# ---------------------------------------------------------------------------
# prepared-weight cache
# ---------------------------------------------------------------------------
W_OIHW, W_OHWI = 0, 1


def weight_layout(t):
    """psg_w_layout of a weight / weight-gradient tensor's MEMORY order: OIHW (torch contiguous) or OHWI
    (torch channels_last, the kernels' native order — optim.ParamArena stores conv weights that way);
    None if it is neither (the caller makes a contiguous copy)."""
    if t.is_contiguous():
        return W_OIHW
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return W_OHWI
    return None


class ParamShadow:
    """Parameter -> (optim.ParamArena, index) for arenas that keep a bf16 shadow copy of the flat master buffer
    (written by the AdamW kernel in its own pass).  For an OHWI / 2-D weight whose K needs no padding the shadow
    slice IS the prepared forward weight, and the data-gradient operand is transposed from it (half the bytes)."""
    _map = {}

    @classmethod
    def register(cls, param, arena, index):
        cls._map[id(param)] = (arena, index, param)

    @classmethod
    def unregister(cls, arena):
        """Drop the entries that belong to `arena` (only those: another arena's parameters keep their shadow)."""
        for k in [k for k, e in cls._map.items() if e[0] is arena]:
            del cls._map[k]

    @classmethod
    def clear(cls):
        cls._map.clear()

    @classmethod
    def lookup(cls, param):
        e = cls._map.get(id(param))
        if e is None or e[2] is not param:
            return None
        arena, index, _ = e
        return arena.shadow_slice(index)        # bf16 1-D tensor in sync with the parameter, or None


class WeightCache:
    """fp32 master weight (OIHW or OHWI memory order) -> kernel-layout weights in the compute dtype.

    wf: [O][Kpad] with k=(kh,kw,ci) for the forward gather, wd: [I][Kpad'] with
    k=(kh,kw,co) for the data-gradient gather (psg_prep_weight).  Entries are
    keyed on the parameter's storage, its torch version counter and a global
    epoch that optimizers writing through raw pointers bump (invalidate()).
    """
    epoch = 0
    _entries = {}

    @classmethod
    def invalidate(cls):
        cls.epoch += 1

    @classmethod
    def clear(cls):
        cls._entries.clear()
        cls.epoch += 1

    @classmethod
    def drop(cls, params):
        """Release the prepared copies of these parameters only (their storage is about to move)."""
        for p in params:
            cls._entries.pop(id(p), None)

    @classmethod
    def get(cls, w, dtype, need_wd):
        key = id(w)
        stamp = (w.data_ptr(), w._version, cls.epoch, dtype)
        ent = cls._entries.get(key)
        if ent is not None and ent[0] == stamp and (ent[2] is not None or not need_wd):
            return ent[1], ent[2]
        lib = _lib_for(w)
        if ent is not None and ent[4] is not None and ent[0][0] == stamp[0] and ent[0][3] == dtype and ent[3] is w:
            # Same parameter, same storage, one optimizer step later, forward operand = the arena's bf16 shadow: everything
            # but the CONTENTS of wd is as it was - the shapes, the shadow view (AdamW rewrote it in place) and the wd buffer,
            # which is refilled in place.  (Rebuilding the entry cost ~25 us of host time per weight and step: 4 ms of a step
            # that is launch-bound below a per-GPU batch of ~64.)
            O, I, ks, code = ent[4]
            shadow = ParamShadow.lookup(w)
            if shadow is not None and shadow.data_ptr() == ent[1].data_ptr():
                wf, wd = ent[1], ent[2]
                if need_wd:
                    if wd is None:
                        wd = torch.empty((I, lib.psg_kpad(ks * ks * O, code)), dtype=dtype, device=w.device)
                    check(lib.psg_prep_weight(ptr(shadow), dtype_code(torch.bfloat16), W_OHWI, None, ptr(wd), O, I, ks, code, stream_ptr()),
                          "psg_prep_weight")
                else:
                    wd = None                      # (stale contents: dropped until a caller needs it again)
                cls._entries[key] = (stamp, wf, wd, w, ent[4])
                return wf, wd
        O, I = w.shape[0], w.shape[1]
        ks = w.shape[2] if w.dim() == 4 else 1
        code = dtype_code(dtype)
        kpf = lib.psg_kpad(ks * ks * I, code)
        kpd = lib.psg_kpad(ks * ks * O, code)
        src = w.detach()
        layout = weight_layout(src)
        shadow = None
        if dtype == torch.bfloat16 and kpf == ks * ks * I and O % 4 == 0 and I % 4 == 0 and (layout == W_OHWI or ks == 1):
            shadow = ParamShadow.lookup(w)
        wd = torch.empty((I, kpd), dtype=dtype, device=w.device) if need_wd else None
        if shadow is not None:                     # the AdamW pass already wrote the forward operand
            wf = shadow.view(O, kpf)
            if need_wd:
                check(lib.psg_prep_weight(ptr(shadow), dtype_code(torch.bfloat16), W_OHWI, None, ptr(wd), O, I, ks, code, stream_ptr()),
                      "psg_prep_weight")
        else:
            wf = torch.empty((O, kpf), dtype=dtype, device=w.device)
            if layout is None:
                src, layout = src.contiguous(), W_OIHW
            check(lib.psg_prep_weight(ptr(src), dtype_code(torch.float32), layout, ptr(wf), ptr(wd), O, I, ks, code, stream_ptr()),
                  "psg_prep_weight")
        # (keep w alive so id() stays unique; the 5th field marks entries whose wf is the shadow: refreshable in place)
        cls._entries[key] = (stamp, wf, wd, w, (O, I, ks, code) if shadow is not None else None)
        return wf, wd


class GradSink:
    """Parameter -> persistent fp32 gradient view (optim.GradArena).  When a parameter is registered, the
    backward kernels write (first use in a step) or accumulate (later uses) its gradient straight into the
    view and autograd gets None: no per-step zero-fill and no AccumulateGrad read-modify-write pass over
    640 M values.  `on_ready(index)` is how the data-parallel reducer learns a gradient is final."""
    _map = {}

    class Entry:
        __slots__ = ("view", "written", "index", "on_ready", "param", "owner")

    @classmethod
    def register(cls, param, view, index, on_ready=None, owner=None):
        """Make `view` the gradient sink of `param`.  A parameter has ONE sink: registering it again from another
        owner (a second GradArena over the same model) displaces the first owner, which is told so
        (`owner.displaced(param)`) and refuses to step from then on instead of silently accumulating through
        autograd's AccumulateGrad into views nothing zeroes."""
        old = cls._map.get(id(param))
        if old is not None and old.param is param and old.owner is not None and old.owner is not owner:
            old.owner.displaced(param)
        e = cls.Entry()
        e.view, e.written, e.index, e.on_ready, e.param, e.owner = view, False, index, on_ready, param, owner
        cls._map[id(param)] = e
        return e

    @classmethod
    def unregister(cls, owner):
        """Remove the sinks registered by `owner` (and only those)."""
        for k in [k for k, e in cls._map.items() if e.owner is owner]:
            del cls._map[k]

    @classmethod
    def unregister_all(cls):
        cls._map.clear()

    @classmethod
    def begin_step(cls, owner=None):
        for e in cls._map.values():
            if owner is None or e.owner is owner:
                e.written = False

    @classmethod
    def get(cls, param):
        e = cls._map.get(id(param))
        return e if (e is not None and e.param is param) else None

    @classmethod
    def done(cls, e):
        e.written = True
        if e.on_ready is not None:
            e.on_ready(e.index)

    @classmethod
    def unwritten(cls, owner=None):
        return [e for e in cls._map.values() if not e.written and (owner is None or e.owner is owner)]


# The two descriptors are packed with struct.pack_into into per-thread buffers (backward runs on autograd's thread) and
# handed over as a pointer: one C call instead of ~30 ctypes field stores (8 us a launch, ~600 launches per train step).
_CONV_FMT = struct.Struct("<13i2fiQ7q8QQq")     # struct psg_conv_desc
_WGRAD_FMT = struct.Struct("<14ifi2q5Qq")       # struct psg_wgrad_desc
assert _CONV_FMT.size == C.sizeof(ConvDesc) and _WGRAD_FMT.size == C.sizeof(WgradDesc)
_tls = threading.local()


def _desc_bufs():
    b = getattr(_tls, "bufs", None)
    if b is None:
        cb, wb = C.create_string_buffer(_CONV_FMT.size), C.create_string_buffer(_WGRAD_FMT.size)
        b = _tls.bufs = (cb, C.cast(cb, C.POINTER(ConvDesc)), wb, C.cast(wb, C.POINTER(WgradDesc)))
    return b


_SPLITK = os.environ.get("PSG_CONV_SPLITK", "1") != "0"
_SPLITK_MAX_OUT = 1 << 23            # M x Cout above which a launch fills the chip anyway


class SplitKStats:
    """How many conv / linear launches were given a split-K workspace (tests, tools)."""
    launches = 0


# psg_conv_fwd_workspace_bytes validates the descriptor and evaluates the tile / split plan: per (geometry, dtype, CUs the
# planner counts on) that answer never changes, and on the launch-bound small-batch / sampler paths the second host call
# per launch was as long as the launch itself
_splitk_need = {}


def _conv_launch(lib, dtype, x, ldx, w, ldw, y, ldy, geom, Cin, Cout, transposed=False, bias=None, rowadd=None,
                 residual=None, ld_res=0, preact=None, dact_u=None, ld_dact=0, act=ACT_NONE, alpha=1.0, drop_p=0.0, seed=0, flags=0):
    B, Hi, Wi, Ho, Wo, ks, stride, pad = geom
    cb, cp, _, _ = _desc_bufs()
    _CONV_FMT.pack_into(cb, 0, dtype_code(dtype), B, Hi, Wi, Cin, Ho, Wo, Cout, ks, stride, pad, int(transposed), act,
                        float(alpha), float(drop_p), int(flags), int(seed) & 0xFFFFFFFFFFFFFFFF,
                        ldx, ldy, rowadd.stride(0) if rowadd is not None else 0, ld_res if residual is not None else 0,
                        preact.stride(-2) if preact is not None else 0, ld_dact if dact_u is not None else 0, ldw,
                        x.data_ptr(), w if isinstance(w, int) else w.data_ptr(), y.data_ptr(),
                        bias.data_ptr() if bias is not None else 0, rowadd.data_ptr() if rowadd is not None else 0,
                        residual.data_ptr() if residual is not None else 0, preact.data_ptr() if preact is not None else 0,
                        dact_u.data_ptr() if dact_u is not None else 0, 0, 0)
    # small grids (sampling, small batches): offer a split-K workspace.  Only launches with few output elements are asked
    # about (one plan evaluation on the host); a batch-256 train step never is.
    if B * Ho * Wo * Cout <= _SPLITK_MAX_OUT and _SPLITK:
        key = (dtype, geom, Cin, Cout, transposed, _lib.AVAIL_CUS[0], _lib.AVAIL_CUS[1])
        need = _splitk_need.get(key)
        if need is None:
            need = _splitk_need[key] = lib.psg_conv_fwd_workspace_bytes(cp)
        if need > 0:
            ws = _lib.workspace(need, x.device)
            struct.pack_into("<Qq", cb, _CONV_FMT.size - 16, ws.data_ptr(), ws.numel())
            SplitKStats.launches += 1
    check(lib.psg_conv_fwd(cp, stream_ptr()), "psg_conv_fwd")


def _wgrad_launch(lib, dtype, x, ldx, dy, lddy, dw, geom, Cin, Cout, accumulate=False, dbias=None, accumulate_bias=False, scale=1.0):
    """dw (+)= dy^T . gather(x); with `dbias` the same launch also produces the bias gradient (column sums of dy)."""
    B, Hi, Wi, Ho, Wo, ks, stride, pad = geom
    layout = weight_layout(dw)
    out = dw
    if layout is None:                       # exotic strides: compute contiguous, copy back
        out, layout = torch.empty(dw.shape, dtype=dw.dtype, device=dw.device), W_OIHW
        if accumulate:
            out.copy_(dw)
    if dbias is not None and (dbias.dtype != torch.float32 or not dbias.is_contiguous() or dbias.numel() != Cout):
        raise _lib.PsgError("wgrad: dbias must be a contiguous fp32 [Cout] tensor")
    _, _, wb, wp = _desc_bufs()

    def pack(ws_ptr, ws_bytes):
        _WGRAD_FMT.pack_into(wb, 0, dtype_code(dtype), B, Hi, Wi, Cin, Ho, Wo, Cout, ks, stride, pad, int(accumulate), layout,
                             int(accumulate_bias) if dbias is not None else 0, float(scale), 0, ldx, lddy,
                             x.data_ptr(), dy.data_ptr(), out.data_ptr(), dbias.data_ptr() if dbias is not None else 0, ws_ptr, ws_bytes)
    pack(0, 0)
    need = lib.psg_conv_wgrad_workspace_bytes(wp)
    if need < 0:
        check(-1, "psg_conv_wgrad_workspace_bytes")
    if need > 0:
        ws = _lib.workspace(need, x.device)
        pack(ws.data_ptr(), ws.numel())
    check(lib.psg_conv_wgrad(wp, stream_ptr()), "psg_conv_wgrad")
    if out is not dw:
        dw.copy_(out)


def _colsum(lib, a, lda, R, groups, cols, dtype, out_dtype, keep2d=False, out=None, accumulate=False):
    if out is None:
        out = torch.empty((groups, cols) if keep2d or groups > 1 else (cols,), dtype=out_dtype, device=a.device)
    need = lib.psg_colsum_workspace_bytes(R, groups, cols)
    ws = _lib.workspace(need, a.device)
    check(lib.psg_colsum(ptr(a), lda, ptr(out), cols, R, groups, cols, dtype_code(dtype), dtype_code(out_dtype), int(accumulate),
                         ptr(ws), ws.numel(), stream_ptr()), "psg_colsum")
    return out


class SideStream:
    """Second HIP stream for the weight-gradient GEMMs of backward: wgrad depends only on the saved input and dY and
    nothing else in backward depends on it, so it fills the tail waves of the data-gradient GEMM that runs next on
    the main stream (-1.5 % step time).  Used only for gradients that go to a GradSink (optim.GradArena): the arena's
    finalize() joins the stream before anything reads the gradients.  OFF by default since round 3 (PSG_WGRAD_STREAM=1
    enables it): it paid -1.5 % in round 1, when the data-gradient kernels left tail waves idle; with the round-3
    weight-gradient tiles two co-running GEMMs only take each other's CUs (same box, batch 256: 82.4 ms with the
    stream, 81.5 without)."""
    enabled = os.environ.get("PSG_WGRAD_STREAM", "0") != "0"
    _streams = {}
    used = False
    # Operands of the launches still in flight on the side stream.  Holding the Python reference does two things:
    # the buffer cannot be recycled, and - the part that matters - autograd's gradient accumulation never adds IN PLACE
    # into a tensor that something else still references (InputBuffer only re-uses a buffer whose use_count is 1).
    # Without it `d_res = dy` (returned by _ConvFn.backward while the side-stream wgrad still reads dy) could be
    # overwritten on the main stream by `dy += other_branch_gradient` under the wgrad's feet.
    _pending = []

    @classmethod
    def get(cls, device):
        k = device.index if device.index is not None else torch.cuda.current_device()
        s = cls._streams.get(k)
        if s is None:
            s = cls._streams[k] = torch.cuda.Stream(device=device)
        return s

    @classmethod
    def hold(cls, *tensors):
        cls._pending.extend(t for t in tensors if t is not None)
        cls.used = True

    @classmethod
    def join(cls, device):
        if cls.used:
            torch.cuda.current_stream(device).wait_stream(cls.get(device))
            cls.used = False
        cls._pending.clear()


def _param_out(param):
    """(out, accumulate, sink entry or None) for a parameter gradient: the registered sink view or a fresh tensor."""
    e = GradSink.get(param)
    if e is not None:
        return e.view, e.written, e
    return torch.empty_like(param), False, None


def _param_ret(out, e):
    """What autograd gets back for a gradient produced into `_param_out`'s tensor."""
    if e is not None:
        GradSink.done(e)
        return None
    return out


def _param_grad(param, compute):
    """Run `compute(out, accumulate)` for a parameter gradient: into the registered sink (returns None for
    autograd) or into a fresh tensor (returned)."""
    e = GradSink.get(param)
    if e is not None:
        compute(e.view, e.written)
        GradSink.done(e)
        return None
    out = torch.empty_like(param)
    compute(out, False)
    return out


# ---------------------------------------------------------------------------
# producing into a slice of a larger buffer (the decoder's concat) and the skip tensors' gradient fan-in
# ---------------------------------------------------------------------------
class SeedSource:
    """The device word every dropout-drawing launch adds to its seed (psg_set_seed_source).  Seeds are launch arguments: a
    train step captured into a hipGraph would replay the same masks forever; with the source enabled the per-site seeds stay
    fixed from step to step (unet._SeedStream restarts its counter each step) and this word, advanced ON THE DEVICE at the end
    of every step, is what changes - eager steps and graph replays then draw identical masks."""
    _t = None
    _users = 0                              # acquire / release count (one per captured train-step graph)
    _STEP = -7046029254386353131            # 0x9E3779B97F4A7C15 as int64 (wraps)

    @classmethod
    def enabled(cls):
        return cls._t is not None

    @classmethod
    def enable(cls, device):
        if cls._t is None:
            lib = _lib.init(device.index if device.index is not None else torch.cuda.current_device())
            cls._t = torch.zeros(1, dtype=torch.int64, device=device)
            check(lib.psg_set_seed_source(ptr(cls._t)), "psg_set_seed_source")
        return cls._t

    @classmethod
    def acquire(cls, device):
        """One count per captured graph: the device word is baked into the captured launches' arguments and must stay alive
        (and keep advancing) until the LAST graph that reads it is closed."""
        cls._users += 1
        return cls.enable(device)

    @classmethod
    def release(cls):
        if cls._users > 0:
            cls._users -= 1
            if cls._users == 0:
                cls.disable()

    @classmethod
    def disable(cls):
        if cls._t is not None:
            check(_lib.load().psg_set_seed_source(None), "psg_set_seed_source")
            cls._t = None
        cls._users = 0

    @classmethod
    def advance(cls):
        if cls._t is not None:
            cls._t.add_(cls._STEP)


class OutSlot:
    """Where an op should produce its result: a row-strided view (same shape as the result) of a larger buffer.  A plain
    Python object on purpose - handed to an autograd Function as a tensor it would count as an INPUT returned as an output."""

    def __init__(self, view):
        self.view = view


def _out_rows(out, shape, dtype, device):
    """(tensor to write, its row stride): a fresh contiguous tensor, or the OutSlot's view after checking that it can be
    addressed as rows (uniform row stride, 16-byte aligned rows)."""
    if out is None:
        return torch.empty(shape, dtype=dtype, device=device), shape[-1]
    v = out.view
    if tuple(v.shape) != tuple(shape) or v.dtype != dtype or v.device != device:
        raise _lib.PsgError(f"out slot {tuple(v.shape)} {v.dtype} does not match the result {tuple(shape)} {dtype}")
    ld = v.stride(-2)
    esz = v.element_size()
    ok = v.stride(-1) == 1 and (v.data_ptr() % 16) == 0 and (ld * esz) % 16 == 0
    expect = ld
    for i in range(v.dim() - 2, -1, -1):
        if v.shape[i] != 1 and v.stride(i) != expect:
            ok = False
        expect *= v.shape[i]
    if not ok:
        raise _lib.PsgError(f"out slot with strides {tuple(v.stride())} is not row-addressable")
    return v, ld


def sum_rows(a, b=None, c=None, out=None):
    """a (+ b (+ c)) over row-strided [..., C] tensors in one pass (psg_sum_rows); `out` may be a strided view."""
    lib = _lib_for(a)
    ar, lda = _rows(a)
    C_ = ar.shape[-1]
    rows = 1
    for n_ in ar.shape[:-1]:
        rows *= int(n_)
    br, ldb = _rows(b) if b is not None else (None, 0)
    cr, ldc = _rows(c) if c is not None else (None, 0)
    if out is None:
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    o, ldo = _out_rows(OutSlot(out), tuple(a.shape), a.dtype, a.device)
    check(lib.psg_sum_rows(ptr(ar), lda, ptr(br) if br is not None else None, ldb, ptr(cr) if cr is not None else None, ldc,
                           ptr(o), ldo, rows, C_, dtype_code(a.dtype), stream_ptr()), "psg_sum_rows")
    return out


class _ConcatSlotFn(torch.autograd.Function):
    """torch.cat([x, skip], dim=-1) where x ALREADY lies in buf[..., :C1] (its producer wrote it there through an OutSlot)
    and skip was copied into buf[..., C1:] when the buffer was made: forward hands out the buffer, backward the two
    slices of its gradient (views: the consumers take row-strided gradients)."""

    @staticmethod
    def forward(ctx, x, skip, holder):
        buf = holder.view
        C1 = x.shape[-1]
        if x.data_ptr() != buf.data_ptr() or x.stride() != buf[..., :C1].stride() or tuple(x.shape[:-1]) != tuple(buf.shape[:-1]):
            raise _lib.PsgError("concat_slot: x was not produced in the buffer's slot")
        ctx.C1 = C1
        return buf

    @staticmethod
    def backward(ctx, dcat):
        return dcat[..., :ctx.C1], dcat[..., ctx.C1:], None


class ConcatSlot:
    """The decoder's concat without the concat: `slot = ConcatSlot(skip, C1)` allocates [.., C1 + C2] and copies the skip
    half (one strided pass, psg_sum_rows); the op producing x gets `out=slot.out`; `slot.cat(x, skip)` is the node whose
    result is the full buffer (reference unet.py:480-504)."""

    def __init__(self, skip, C1):
        shape = tuple(skip.shape[:-1]) + (C1 + skip.shape[-1],)
        self.buf = torch.empty(shape, dtype=skip.dtype, device=skip.device)
        with torch.no_grad():
            sum_rows(skip.detach(), out=self.buf[..., C1:])
        self.out = OutSlot(self.buf[..., :C1])

    def cat(self, x, skip):
        return _ConcatSlotFn.apply(x, skip, OutSlot(self.buf))


class _Fan3Fn(torch.autograd.Function):
    """One tensor, three consumers: the three gradients are summed in ONE pass (fp32 sum, one rounding) instead of
    autograd's clone of the first (a strided concat-gradient slice) plus two adds."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        gs = [g for g in (g0, g1, g2) if g is not None]
        if not gs:
            return None
        if len(gs) == 1:
            return gs[0]
        return sum_rows(gs[0], gs[1], gs[2] if len(gs) > 2 else None)


def fan3(x):
    """(x, x, x) for a skip tensor's three consumers (next encoder stage, two decoder blocks); see _Fan3Fn."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x, x
    return _Fan3Fn.apply(x)


# ---------------------------------------------------------------------------
# conv / linear
# ---------------------------------------------------------------------------
class _ConvFn(torch.autograd.Function):
    """y = residual + alpha * drop(act(conv(x, W) + bias + rowadd[b]))  — psg_conv_fwd / psg_conv_wgrad."""

    @staticmethod
    def forward(ctx, x, weight, bias, rowadd, residual, stride, act, alpha, drop_p, seed, out=None):
        lib = _lib_for(x)
        dtype = x.dtype
        is_conv = weight.dim() == 4
        Cout, Cin = weight.shape[0], weight.shape[1]
        ks = weight.shape[2] if is_conv else 1
        pad = 1 if ks == 3 else 0
        xr, ldx = _rows(x)
        if is_conv:
            B, Hi, Wi = x.shape[0], x.shape[1], x.shape[2]
            Ho, Wo = (Hi + 2 * pad - ks) // stride + 1, (Wi + 2 * pad - ks) // stride + 1
            out_shape = (B, Ho, Wo, Cout)
        else:
            B, Hi, Wi, Ho, Wo = xr.numel() // xr.shape[-1], 1, 1, 1, 1
            out_shape = tuple(x.shape[:-1]) + (Cout,)
        geom = (B, Hi, Wi, Ho, Wo, ks, stride, pad)
        need_dx = ctx.needs_input_grad[0]
        wf, wd = WeightCache.get(weight, dtype, need_dx)
        y, ldy = _out_rows(out, out_shape, dtype, x.device)
        has_epi = (act != ACT_NONE)
        any_grad = any(ctx.needs_input_grad)
        preact = torch.empty(out_shape, dtype=dtype, device=x.device) if (has_epi and any_grad) else None
        res_r, ld_res = (None, 0)
        if residual is not None:
            res_r, ld_res = _rows(residual)
        ra = None
        if rowadd is not None:
            ra = rowadd if (rowadd.stride(-1) == 1 and rowadd.stride(0) % 4 == 0 and rowadd.data_ptr() % 16 == 0) else rowadd.contiguous()
        _conv_launch(lib, dtype, xr, ldx, wf, 0, y, ldy, geom, Cin, Cout, bias=bias, rowadd=ra, residual=res_r, ld_res=ld_res,
                     preact=preact, act=act, alpha=alpha, drop_p=drop_p, seed=seed)
        ctx.save_for_backward(xr, weight, preact)
        ctx.bias_param, ctx.weight_param = bias, weight
        ctx.meta = (geom, Cin, Cout, ldx, act, alpha, drop_p, seed, bias is not None, rowadd is not None, residual is not None, tuple(x.shape), wd)
        return y

    @staticmethod
    def backward(ctx, dy):
        xr, weight, preact = ctx.saved_tensors
        geom, Cin, Cout, ldx, act, alpha, drop_p, seed, has_bias, has_ra, has_res, x_shape, wd = ctx.meta
        B, Hi, Wi, Ho, Wo, ks, stride, pad = geom
        lib = _lib_for(dy)
        dtype = dy.dtype
        dyr, lddy = _rows(dy)
        M = B * Ho * Wo
        d_res = dy if has_res else None
        # gradient w.r.t. the pre-epilogue accumulator; a pure output gate (alpha only) is folded into the dgrad
        # epilogue and the wgrad scale instead of a pass over dy
        gate = 1.0
        if act == ACT_NONE and drop_p == 0.0 and alpha != 1.0 and not has_ra:
            gate = float(alpha)
            g, ldg = dyr, lddy
        elif act != ACT_NONE or drop_p > 0.0 or alpha != 1.0:
            g = torch.empty((M, Cout), dtype=dtype, device=dy.device)
            check(lib.psg_epilogue_bwd(ptr(dyr), lddy, ptr(preact), Cout, ptr(g), Cout, M, Cout, act, float(alpha), float(drop_p),
                                       int(seed), dtype_code(dtype), stream_ptr()), "psg_epilogue_bwd")
            ldg = Cout
        else:
            g, ldg = dyr, lddy
        dx = dw = db = dra = None
        if ctx.needs_input_grad[0]:
            if wd is None:
                _, wd = WeightCache.get(weight, dtype, True)
            dx = torch.empty(x_shape, dtype=dtype, device=dy.device)
            tgeom = (B, Ho, Wo, Hi, Wi, ks, stride, pad)     # gather source = dY grid, result = input grid
            _conv_launch(lib, dtype, g, ldg, wd, 0, dx, Cin, tgeom, Cout, Cin, transposed=True, alpha=gate)
        want_b = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            # one launch: weight gradient + (fused) bias gradient, straight into the gradient arena when registered
            wo, wacc, we = _param_out(ctx.weight_param)
            bo, bacc, be = _param_out(ctx.bias_param) if want_b else (None, False, None)
            if SideStream.enabled and we is not None:
                side, cur = SideStream.get(dy.device), torch.cuda.current_stream(dy.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    _wgrad_launch(lib, dtype, xr, ldx, g, ldg, wo, geom, Cin, Cout, accumulate=wacc, dbias=bo, accumulate_bias=bacc, scale=gate)
                g.record_stream(side); xr.record_stream(side)
                SideStream.hold(g, xr, dy)
            else:
                _wgrad_launch(lib, dtype, xr, ldx, g, ldg, wo, geom, Cin, Cout, accumulate=wacc, dbias=bo, accumulate_bias=bacc, scale=gate)
            dw = _param_ret(wo, we)
            if want_b:
                db = _param_ret(bo, be)
        elif want_b:
            def bias_only(out, acc):
                if gate == 1.0:
                    _colsum(lib, g, ldg, M, 1, Cout, dtype, torch.float32, out=out, accumulate=acc)
                else:                      # (rare: frozen weight, trainable bias, gated output)
                    t = _colsum(lib, g, ldg, M, 1, Cout, dtype, torch.float32).mul_(gate)
                    out.add_(t) if acc else out.copy_(t)
            db = _param_grad(ctx.bias_param, bias_only)
        if has_ra and ctx.needs_input_grad[3]:
            dra = _colsum(lib, g, ldg, Ho * Wo, B, Cout, dtype, dtype, keep2d=True)
        return dx, dw, db, dra, d_res, None, None, None, None, None, None


def conv2d(x, weight, bias=None, stride=1, rowadd=None, residual=None, act=ACT_NONE, alpha=1.0, drop_p=0.0, seed=0, out=None):
    """x: [B,H,W,Cin] channels-last; weight: fp32 OIHW parameter (3x3 pad 1, or 1x1).  nn.Conv2d of unet.py.
    `out`: a row-strided tensor of the output's shape to produce the result in (`OutSlot`, see `concat_slot`)."""
    return _ConvFn.apply(x, weight, bias, rowadd, residual, stride, act, alpha, drop_p, seed, out)


def linear(x, weight, bias=None, residual=None, act=ACT_NONE, alpha=1.0, drop_p=0.0, seed=0, out=None):
    """x: [..., Cin]; weight: fp32 [Cout, Cin] parameter.  nn.Linear of unet.py, with the fused epilogue."""
    return _ConvFn.apply(x, weight, bias, None, residual, 1, act, alpha, drop_p, seed, out)


_FFN_SAVE_DACT = os.environ.get("PSG_FFN_SAVE_DACT", "1") != "0"     # 0: save u, re-evaluate gelu'(u) and the mask in backward (A/B)
_GN_SPLIT = os.environ.get("PSG_GN_SPLIT", "1") != "0"               # 0: plain GroupNorm node, autograd adds the bypass gradient (A/B)


class _FFNFn(torch.autograd.Function):
    """y = x + alpha * drop2(W2 . drop1(gelu(W1 x + b1)) + b2)   - the FFN of CrossAttentionBlock (unet.py:176-187, 250).

    One node instead of two `_ConvFn`s so that backward can use the kernels' fused forms: the data gradient of the
    second Linear is produced ALREADY multiplied by GELU'(u) and the first dropout mask (psg_conv_fwd's dact_u
    form: no pass over the [M, hidden] gradient), and the residual branch's gradient is added in the epilogue of the
    first Linear's data gradient (no autograd add)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, alpha, drop_p, seed1, seed2, out=None):
        lib = _lib_for(x)
        dtype = x.dtype
        C, Hd = w1.shape[1], w1.shape[0]
        xr, ldx = _rows(x)
        M = xr.numel() // C
        need = any(ctx.needs_input_grad)
        wf1, wd1 = WeightCache.get(w1, dtype, need)
        wf2, wd2 = WeightCache.get(w2, dtype, need)
        u = torch.empty((M, Hd), dtype=dtype, device=x.device) if need else None
        hmid = torch.empty((M, Hd), dtype=dtype, device=x.device)
        y, ldy = _out_rows(out, tuple(x.shape), dtype, x.device)
        g1 = (M, 1, 1, 1, 1, 1, 1, 0)
        # `u` receives gelu'(W1 x + b1) * mask1 / (1 - p), not the pre-activation: backward multiplies by it
        _conv_launch(lib, dtype, xr, ldx, wf1, 0, hmid, Hd, g1, C, Hd, bias=b1, preact=u, act=ACT_GELU, drop_p=drop_p, seed=seed1,
                     flags=_lib.CONV_SAVE_DACT if (u is not None and _FFN_SAVE_DACT) else 0)
        _conv_launch(lib, dtype, hmid, Hd, wf2, 0, y, ldy, g1, Hd, C, bias=b2, residual=xr, ld_res=ldx, alpha=alpha, drop_p=drop_p, seed=seed2)
        ctx.save_for_backward(xr, u, hmid, w1, w2)
        ctx.params = (w1, b1, w2, b2)
        ctx.meta = (M, C, Hd, ldx, alpha, drop_p, seed1, seed2, tuple(x.shape), wd1, wd2)
        return y

    @staticmethod
    def backward(ctx, dy):
        xr, u, hmid, w1, w2 = ctx.saved_tensors
        w1p, b1p, w2p, b2p = ctx.params
        M, C, Hd, ldx, alpha, drop_p, seed1, seed2, x_shape, wd1, wd2 = ctx.meta
        lib = _lib_for(dy)
        dtype = dy.dtype
        dyr, lddy = _rows(dy)
        geo = (M, 1, 1, 1, 1, 1, 1, 0)
        if wd1 is None:
            _, wd1 = WeightCache.get(w1, dtype, True)
        if wd2 is None:
            _, wd2 = WeightCache.get(w2, dtype, True)
        # gradient w.r.t. the second Linear's accumulator (dropout 2 and the 0.6 gate)
        g2 = torch.empty((M, C), dtype=dtype, device=dy.device)
        check(lib.psg_epilogue_bwd(ptr(dyr), lddy, None, C, ptr(g2), C, M, C, ACT_NONE, float(alpha), float(drop_p), int(seed2),
                                   dtype_code(dtype), stream_ptr()), "psg_epilogue_bwd")
        # d/du of the first Linear's pre-activation: dgrad of Linear 2 with the backward-form epilogue
        gu = torch.empty((M, Hd), dtype=dtype, device=dy.device)
        if _FFN_SAVE_DACT:
            _conv_launch(lib, dtype, g2, C, wd2, 0, gu, Hd, geo, C, Hd, transposed=True, dact_u=u, ld_dact=Hd, flags=_lib.CONV_DACT_MUL)
        else:
            _conv_launch(lib, dtype, g2, C, wd2, 0, gu, Hd, geo, C, Hd, transposed=True, dact_u=u, ld_dact=Hd, act=ACT_GELU, drop_p=drop_p, seed=seed1)

        def wgrad(xin, ldxin, g, ldg, wp, bp, cin, cout):
            wo, wacc, we = _param_out(wp)
            bo, bacc, be = _param_out(bp)
            if SideStream.enabled and we is not None:
                side, cur = SideStream.get(dy.device), torch.cuda.current_stream(dy.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    _wgrad_launch(lib, dtype, xin, ldxin, g, ldg, wo, geo, cin, cout, accumulate=wacc, dbias=bo, accumulate_bias=bacc)
                g.record_stream(side); xin.record_stream(side)
                SideStream.hold(g, xin)
            else:
                _wgrad_launch(lib, dtype, xin, ldxin, g, ldg, wo, geo, cin, cout, accumulate=wacc, dbias=bo, accumulate_bias=bacc)
            return _param_ret(wo, we), _param_ret(bo, be)

        dw2 = db2 = dw1 = db1 = dx = None
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            dw2, db2 = wgrad(hmid, Hd, g2, C, w2p, b2p, Hd, C)
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x_shape, dtype=dtype, device=dy.device)
            # dx = W1^T gu + dy (the residual branch's gradient rides in the epilogue)
            _conv_launch(lib, dtype, gu, Hd, wd1, 0, dx, C, geo, Hd, C, transposed=True, residual=dyr, ld_res=lddy)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw1, db1 = wgrad(xr, ldx, gu, Hd, w1p, b1p, C, Hd)
        return dx, dw1, db1, dw2, db2, None, None, None, None, None


def ffn(x, w1, b1, w2, b2, alpha, drop_p=0.0, seed1=0, seed2=0, out=None):
    """x + alpha * drop(W2 drop(gelu(W1 x + b1)) + b2), fused forward epilogues and backward forms (see _FFNFn)."""
    return _FFNFn.apply(x, w1, b1, w2, b2, alpha, drop_p, seed1, seed2, out)


class _CrossInProjFn(torch.autograd.Function):
    """Packed MHA in-projection for cross-attention (unet.py:235 -> F.multi_head_attention_forward):
    q = xn @ W[:E]^T + b[:E];  kv = tp @ W[E:]^T + b[E:]  with ONE packed parameter [3E, E]."""

    @staticmethod
    def forward(ctx, xn, tp, weight, bias):
        lib = _lib_for(xn)
        dtype = xn.dtype
        E = weight.shape[1]
        wf, wd = WeightCache.get(weight, dtype, ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        xr, ldx = _rows(xn)
        tr, ldt = _rows(tp)
        Mq, Mk = xn.numel() // E, tp.numel() // E
        q = torch.empty(tuple(xn.shape[:-1]) + (E,), dtype=dtype, device=xn.device)
        kv = torch.empty(tuple(tp.shape[:-1]) + (2 * E,), dtype=dtype, device=xn.device)
        kp = wf.shape[1]
        esz = wf.element_size()
        _conv_launch(lib, dtype, xr, ldx, wf.data_ptr(), 0, q, E, (Mq, 1, 1, 1, 1, 1, 1, 0), E, E, bias=bias[:E])
        _conv_launch(lib, dtype, tr, ldt, wf.data_ptr() + E * kp * esz, 0, kv, 2 * E, (Mk, 1, 1, 1, 1, 1, 1, 0), E, 2 * E, bias=bias[E:])
        ctx.save_for_backward(xr, tr, weight)
        ctx.bias_param, ctx.weight_param = bias, weight
        ctx.meta = (E, ldx, ldt, Mq, Mk, tuple(xn.shape), tuple(tp.shape), wd, wf)
        return q, kv

    @staticmethod
    def backward(ctx, dq, dkv):
        xr, tr, weight = ctx.saved_tensors
        E, ldx, ldt, Mq, Mk, xs, ts, wd, wf = ctx.meta
        lib = _lib_for(dq)
        dtype = dq.dtype
        dqr, lddq = _rows(dq)
        dkr, lddk = _rows(dkv)
        if wd is None:
            _, wd = WeightCache.get(weight, dtype, True)
        kpd = wd.shape[1]
        esz = wd.element_size()
        dxn = dtp = dw = db = None
        if ctx.needs_input_grad[0]:
            dxn = torch.empty(xs, dtype=dtype, device=dq.device)
            _conv_launch(lib, dtype, dqr, lddq, wd.data_ptr(), kpd, dxn, E, (Mq, 1, 1, 1, 1, 1, 1, 0), E, E, transposed=True)
        if ctx.needs_input_grad[1]:
            dtp = torch.empty(ts, dtype=dtype, device=dq.device)
            _conv_launch(lib, dtype, dkr, lddk, wd.data_ptr() + E * esz, kpd, dtp, E, (Mk, 1, 1, 1, 1, 1, 1, 0), 2 * E, E, transposed=True)
        if ctx.needs_input_grad[2]:
            wo, wacc, we = _param_out(ctx.weight_param)
            bo, bacc, be = _param_out(ctx.bias_param) if ctx.needs_input_grad[3] else (None, False, None)
            _wgrad_launch(lib, dtype, xr, ldx, dqr, lddq, wo[:E], (Mq, 1, 1, 1, 1, 1, 1, 0), E, E, accumulate=wacc,
                          dbias=None if bo is None else bo[:E], accumulate_bias=bacc)
            _wgrad_launch(lib, dtype, tr, ldt, dkr, lddk, wo[E:], (Mk, 1, 1, 1, 1, 1, 1, 0), E, 2 * E, accumulate=wacc,
                          dbias=None if bo is None else bo[E:], accumulate_bias=bacc)
            dw = _param_ret(wo, we)
            if bo is not None:
                db = _param_ret(bo, be)
        elif ctx.needs_input_grad[3]:
            def bg(out, acc):
                _colsum(lib, dqr, lddq, Mq, 1, E, dtype, torch.float32, out=out[:E], accumulate=acc)
                _colsum(lib, dkr, lddk, Mk, 1, 2 * E, dtype, torch.float32, out=out[E:], accumulate=acc)
            db = _param_grad(ctx.bias_param, bg)
        return dxn, dtp, dw, db


def cross_in_proj(xn, tp, weight, bias):
    return _CrossInProjFn.apply(xn, tp, weight, bias)


# ---------------------------------------------------------------------------
# the 17 ResBlocks' time_proj / text_proj as TWO GEMMs (unet.py:83-86,119-124)
# ---------------------------------------------------------------------------
class _SplitColsFn(torch.autograd.Function):
    """[B, sum C] -> 17 column slices (views); backward = ONE concatenation of the 17 gradients (autograd's own slice nodes
    would each allocate and fill a zero [B, sum C] tensor)."""

    @staticmethod
    def forward(ctx, x, bounds):
        ctx.bounds, ctx.shape = bounds, tuple(x.shape)
        return tuple(x[:, a:b] for a, b in bounds)

    @staticmethod
    def backward(ctx, *grads):
        if any(g is None for g in grads):
            full = torch.zeros(ctx.shape, dtype=next(g for g in grads if g is not None).dtype, device=next(g for g in grads if g is not None).device)
            for (a, b), g in zip(ctx.bounds, grads):
                if g is not None:
                    full[:, a:b] = g
            return full, None
        return torch.cat(grads, dim=1), None


class ProjGroup:
    """`h += time_proj(temb) + text_proj(pooled)` of every ResBlock (unet.py:119-124) depends only on temb / pooled: with the 17
    weights of a kind ADJACENT in the parameter arena (UNet.arena_layout) they are one [sum Cout, K] operand, and the 34 tiny
    M = batch GEMMs of a forward (+ 34 data gradients, 34 weight gradients and the 32 accumulation adds of temb's / pooled's
    gradients in backward) become two Linear nodes over VIRTUAL parameters: leaf views of the flat master buffer whose
    gradient sink is the matching run of the gradient arena and whose prepared bf16 operand is the matching run of the AdamW
    shadow.  state_dict, the optimizer and the all-reduce never see them - they see the 68 real parameters, which are marked
    written (and reported to the data-parallel reducer) when the group's gradient lands."""

    def __init__(self, blocks, param_arena, grad_arena):
        self.ok = False
        self.blocks = list(blocks)
        pidx = {id(p): i for i, p in enumerate(param_arena.params)}
        self.pa, self.ga = param_arena, grad_arena
        self.virtual = []                       # (w_all, b_all) per kind
        self.bounds, off = [], 0
        for blk in self.blocks:
            self.bounds.append((off, off + blk.out_channels))
            off += blk.out_channels
        self.width = off
        self.members = []
        for kind in ("time_proj", "text_proj"):
            ws, bs = [getattr(b, kind).weight for b in self.blocks], [getattr(b, kind).bias for b in self.blocks]
            if any(id(t) not in pidx for t in ws + bs):
                return
            wi, bi = tuple(pidx[id(t)] for t in ws), tuple(pidx[id(t)] for t in bs)
            for idx in (wi, bi):                # one contiguous run each, in block order, in BOTH arenas
                for a, b in zip(idx[:-1], idx[1:]):
                    if param_arena.offsets[b] != param_arena.offsets[a] + param_arena.params[a].numel() or grad_arena.offsets[b] != param_arena.offsets[b]:
                        return
            K = ws[0].shape[1]
            ow, ob = param_arena.offsets[wi[0]], param_arena.offsets[bi[0]]
            w_all = param_arena.flat[ow:ow + self.width * K].view(self.width, K).requires_grad_(True)
            b_all = param_arena.flat[ob:ob + self.width].requires_grad_(True)
            gw = grad_arena.flat[ow:ow + self.width * K].view(self.width, K)
            gb = grad_arena.flat[ob:ob + self.width]
            for virt, view, idx in ((w_all, gw, wi), (b_all, gb, bi)):
                ents = [grad_arena.entries[i] for i in idx]
                ve = GradSink.register(virt, view, -1, on_ready=(lambda _i, ents=ents: [GradSink.done(e) for e in ents]), owner=grad_arena)
                grad_arena.extra_entries.append(ve)
                ParamShadow.register(virt, param_arena, idx)
            self.virtual.append((w_all, b_all))
            self.members.append(ws + bs)
        self._versions = self._member_versions()
        self.ok = True

    def _member_versions(self):
        return tuple(p._version for m in self.members for p in m)

    def usable(self):
        """Both arenas still own the parameters, and a member changed behind the optimizer's back (load_state_dict, an in-place
        torch op: its own version counter, which the virtual views do not share) drops the group's prepared operands."""
        if not self.ok or self.pa._displaced_by is not None or self.ga._displaced:
            return False
        v = self._member_versions()
        if v != self._versions:
            self._versions = v
            WeightCache.drop([t for pair in self.virtual for t in pair])
        return True

    def release(self):
        WeightCache.drop([t for pair in self.virtual for t in pair])
        for pair in self.virtual:
            for t in pair:
                e = GradSink._map.pop(id(t), None)
                if e is not None and e in self.ga.extra_entries:
                    self.ga.extra_entries.remove(e)
                ParamShadow._map.pop(id(t), None)
        self.ok = False

    def rowadds(self, temb, pooled):
        """The 17 per-sample additive vectors [B, Cout_k] (views of one [B, sum Cout] result)."""
        (wt, bt), (wx, bx) = self.virtual
        ra = linear(temb, wt, bt)
        ra = linear(pooled, wx, bx, residual=ra)
        if torch.is_grad_enabled() and ra.requires_grad:
            return _SplitColsFn.apply(ra, tuple(self.bounds))
        return tuple(ra[:, a:b] for a, b in self.bounds)


# ---------------------------------------------------------------------------
# GroupNorm (+SiLU)
# ---------------------------------------------------------------------------
class _GroupNormFn(torch.autograd.Function):
    """y = GroupNorm(x) [+SiLU]; with `passthrough` the node has a second output that IS x (for the consumer that
    bypasses the norm - ResBlock skip, attention residual): backward then receives both gradients and adds the
    bypass one inside the GroupNorm-backward kernel (psg_groupnorm_bwd_res) instead of autograd running a separate
    accumulation pass over the activation gradient."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu, passthrough=False):
        lib = _lib_for(x)
        dtype = x.dtype
        xr, ldx = _rows(x)
        B, Cc = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * Cc)
        y = torch.empty(x.shape, dtype=dtype, device=x.device)
        stats = torch.empty((2, B * groups), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(lib.psg_groupnorm_fwd_workspace_bytes(B, groups), x.device)
        check(lib.psg_groupnorm_fwd(ptr(xr), ldx, ptr(y), Cc, ptr(gamma), ptr(beta), ptr(stats[0]), ptr(stats[1]), B, HW, Cc, groups,
                                    float(eps), int(silu), dtype_code(dtype), ptr(ws), stream_ptr()), "psg_groupnorm_fwd")
        ctx.save_for_backward(xr, gamma, beta, stats)
        ctx.gamma_param, ctx.beta_param = gamma, beta
        ctx.meta = (B, HW, Cc, groups, silu, ldx, tuple(x.shape))
        if passthrough:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        xr, gamma, beta, stats = ctx.saved_tensors
        B, HW, Cc, groups, silu, ldx, shape = ctx.meta
        if dy is None:                               # only the bypass consumer produced a gradient
            return dpass, None, None, None, None, None, None
        lib = _lib_for(dy)
        dtype = dy.dtype
        dyr, lddy = _rows(dy)
        dres, lddres = (None, 0)
        if dpass is not None:
            dres, lddres = _rows(dpass if dpass.dtype == dtype else dpass.to(dtype))
        dx = torch.empty(shape, dtype=dtype, device=dy.device)
        eg, eb = GradSink.get(ctx.gamma_param), GradSink.get(ctx.beta_param)
        sink = eg is not None and eb is not None and eg.written == eb.written
        if sink:
            dg, db, acc = eg.view, eb.view, eg.written
        else:
            dg = torch.empty(Cc, dtype=torch.float32, device=dy.device)
            db = torch.empty(Cc, dtype=torch.float32, device=dy.device)
            acc = False
        ws = _lib.workspace(lib.psg_groupnorm_bwd_workspace_bytes(B, Cc), dy.device)
        check(lib.psg_groupnorm_bwd_res(ptr(dyr), lddy, ptr(xr), ldx, ptr(gamma), ptr(beta), ptr(stats[0]), ptr(stats[1]), ptr(dres), lddres,
                                        ptr(dx), Cc, ptr(dg), ptr(db), B, HW, Cc, groups, int(silu), int(acc), dtype_code(dtype), ptr(ws),
                                        stream_ptr()), "psg_groupnorm_bwd_res")
        if sink:
            GradSink.done(eg)
            GradSink.done(eb)
            return dx, None, None, None, None, None, None
        return dx, dg, db, None, None, None, None


def group_norm(x, gamma, beta, groups, eps=1e-5, silu=False):
    """nn.GroupNorm (+F.silu) on channels-last [B, ..., C]: unet.py:115,127,214,231,397."""
    return _GroupNormFn.apply(x, gamma, beta, groups, eps, silu)


def group_norm_split(x, gamma, beta, groups, eps=1e-5, silu=False):
    """(GroupNorm(x), x): use the second result wherever x itself is consumed next to the norm (skip / residual
    paths), so the two gradients meet inside the GroupNorm-backward kernel (see _GroupNormFn)."""
    if not (_GN_SPLIT and torch.is_grad_enabled() and x.requires_grad):
        return _GroupNormFn.apply(x, gamma, beta, groups, eps, silu), x
    return _GroupNormFn.apply(x, gamma, beta, groups, eps, silu, True)


# ---------------------------------------------------------------------------
# attention core
# ---------------------------------------------------------------------------
class _AttnFn(torch.autograd.Function):
    """softmax((q/sqrt(d)) k^T) v for packed projections.  self: qkv [B,L,3E]; cross: q [B,L,E], kv [B,S,2E]."""

    @staticmethod
    def forward(ctx, q_src, kv_src, heads, drop_p, seed):
        lib = _lib_for(q_src)
        dtype = q_src.dtype
        self_mode = kv_src is None
        qr, ldq = _rows(q_src)
        B, L = q_src.shape[0], q_src.shape[1]
        esz = qr.element_size()
        if self_mode:
            E = q_src.shape[-1] // 3
            S = L
            kp, vp, ldk = qr.data_ptr() + E * esz, qr.data_ptr() + 2 * E * esz, ldq
            kvr = None
        else:
            E = q_src.shape[-1]
            kvr, ldk = _rows(kv_src)
            S = kv_src.shape[1]
            kp, vp = kvr.data_ptr(), kvr.data_ptr() + E * esz
        d = E // heads
        scale = float(d) ** -0.5
        o = torch.empty((B, L, E), dtype=dtype, device=q_src.device)
        lse = torch.empty((B, heads, L), dtype=torch.float32, device=q_src.device)
        check(lib.psg_attn_fwd(qr.data_ptr(), ldq, kp, ldk, vp, ldk, ptr(o), E, ptr(lse), B, heads, L, S, d, scale, float(drop_p),
                               int(seed), dtype_code(dtype), stream_ptr()), "psg_attn_fwd")
        ctx.save_for_backward(qr, kvr, o, lse)
        ctx.meta = (self_mode, B, L, S, E, heads, d, scale, drop_p, seed, ldq, ldk, tuple(q_src.shape), None if self_mode else tuple(kv_src.shape))
        return o

    @staticmethod
    def backward(ctx, do):
        qr, kvr, o, lse = ctx.saved_tensors
        self_mode, B, L, S, E, heads, d, scale, drop_p, seed, ldq, ldk, qshape, kvshape = ctx.meta
        lib = _lib_for(do)
        dtype = do.dtype
        dor, lddo = _rows(do)
        esz = qr.element_size()
        delta = torch.empty((B, heads, L), dtype=torch.float32, device=do.device)
        if self_mode:
            dqkv = torch.empty(qshape, dtype=dtype, device=do.device)
            W3 = 3 * E
            kp, vp = qr.data_ptr() + E * esz, qr.data_ptr() + 2 * E * esz
            dqp, dkp, dvp, lddq, lddk = dqkv.data_ptr(), dqkv.data_ptr() + E * esz, dqkv.data_ptr() + 2 * E * esz, W3, W3
            dkv = None
        else:
            dqkv = torch.empty(qshape, dtype=dtype, device=do.device)
            dkv = torch.empty(kvshape, dtype=dtype, device=do.device)
            kp, vp = kvr.data_ptr(), kvr.data_ptr() + E * esz
            dqp, dkp, dvp, lddq, lddk = dqkv.data_ptr(), dkv.data_ptr(), dkv.data_ptr() + E * esz, E, 2 * E
        check(lib.psg_attn_bwd(qr.data_ptr(), ldq, kp, ldk, vp, ldk, ptr(o), E, ptr(dor), lddo, ptr(lse), ptr(delta), dqp, lddq, dkp,
                               lddk, dvp, lddk, B, heads, L, S, d, scale, float(drop_p), int(seed), dtype_code(dtype), stream_ptr()),
              "psg_attn_bwd")
        return dqkv, dkv, None, None, None


def attention_self(qkv, heads, drop_p=0.0, seed=0):
    return _AttnFn.apply(qkv, None, heads, drop_p, seed)


def attention_cross(q, kv, heads, drop_p=0.0, seed=0):
    return _AttnFn.apply(q, kv, heads, drop_p, seed)


# ---------------------------------------------------------------------------
# small ops
# ---------------------------------------------------------------------------
class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        lib = _lib_for(x)
        xr, ldx = _rows(x)
        B, Hi, Wi, Cc = x.shape
        y = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
        check(lib.psg_upsample_bilinear_fwd(ptr(xr), ldx, ptr(y), Cc, B, Hi, Wi, Ho, Wo, Cc, dtype_code(x.dtype), stream_ptr()),
              "psg_upsample_bilinear_fwd")
        ctx.meta = (B, Hi, Wi, Ho, Wo, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Hi, Wi, Ho, Wo, Cc = ctx.meta
        lib = _lib_for(dy)
        dyr, lddy = _rows(dy)
        dx = torch.empty((B, Hi, Wi, Cc), dtype=dy.dtype, device=dy.device)
        check(lib.psg_upsample_bilinear_bwd(ptr(dyr), lddy, ptr(dx), Cc, B, Hi, Wi, Ho, Wo, Cc, dtype_code(dy.dtype), stream_ptr()),
              "psg_upsample_bilinear_bwd")
        return dx, None, None


def upsample_bilinear(x, size):
    """nn.Upsample(size, mode='bilinear', align_corners=False) — unet.py:365,375,385."""
    return _UpsampleFn.apply(x, int(size[0]), int(size[1]))


class _ToNCHWFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib_for(x)
        xr, ldx = _rows(x)
        B, H, W, Cc = x.shape
        y = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
        check(lib.psg_nhwc_to_nchw(ptr(xr), ldx, ptr(y), B, Cc, H * W, dtype_code(x.dtype), stream_ptr()), "psg_nhwc_to_nchw")
        ctx.meta = (B, H, W, Cc, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, Cc, dtype = ctx.meta
        lib = _lib_for(dy)
        dyc = dy.contiguous().float()
        dx = torch.empty((B, H, W, Cc), dtype=dtype, device=dy.device)
        check(lib.psg_nchw_to_nhwc(ptr(dyc), ptr(dx), Cc, B, Cc, H * W, dtype_code(dtype), stream_ptr()), "psg_nchw_to_nhwc")
        return dx


def nhwc_to_nchw(x):
    """channels-last compute dtype -> the reference's NCHW fp32 boundary layout (differentiable)."""
    return _ToNCHWFn.apply(x)


def nchw_to_nhwc(x, dtype):
    """[B,C,H,W] fp32 -> [B,H,W,C] compute dtype (network input; no gradient)."""
    lib = _lib_for(x)
    xc = x.detach().contiguous().float()
    B, Cc, H, W = xc.shape
    y = torch.empty((B, H, W, Cc), dtype=dtype, device=x.device)
    check(lib.psg_nchw_to_nhwc(ptr(xc), ptr(y), Cc, B, Cc, H * W, dtype_code(dtype), stream_ptr()), "psg_nchw_to_nhwc")
    return y


def text_pool(text, dtype):
    """AdaptiveAvgPool1d(1) over tokens (unet.py:445) + compute-dtype copy of the tokens."""
    lib = _lib_for(text)
    tc = text.detach().contiguous().float()
    B, S, D = tc.shape
    pooled = torch.empty((B, D), dtype=dtype, device=text.device)
    cast = torch.empty((B, S, D), dtype=dtype, device=text.device)
    check(lib.psg_text_pool(ptr(tc), ptr(pooled), D, ptr(cast), B, S, D, dtype_code(dtype), stream_ptr()), "psg_text_pool")
    return pooled, cast


def timestep_sinusoid(t, coeff, dtype):
    """cat[sin(t*coeff), cos(t*coeff)] (unet.py:47-50)."""
    lib = _lib_for(coeff)
    tt = t.detach().to(device=coeff.device, dtype=torch.int64).contiguous()
    B, half = tt.shape[0], coeff.shape[0]
    out = torch.empty((B, 2 * half), dtype=dtype, device=coeff.device)
    check(lib.psg_timestep_sinusoid(ptr(tt), ptr(coeff.detach().float().contiguous()), ptr(out), 2 * half, B, half, dtype_code(dtype),
                                    stream_ptr()), "psg_timestep_sinusoid")
    return out

"""ctypes binding of libpsg_hip.so (the C ABI declared in include/psg_hip.h).

The product path has NO fallback: if the shared library is missing, any attempt
to run an op raises (``load()``), it never reroutes to a CPU or eager path.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSG_LIB_PATH") or os.path.join(_HERE, "libpsg_hip.so")   # override: A/B kernel builds in one GPU session

PSG_F32, PSG_BF16 = 0, 1
ACT_NONE, ACT_SILU, ACT_GELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3, 4
CONV_SAVE_DACT, CONV_DACT_MUL, CONV_GENERIC_EPILOGUE = 1, 2, 4          # enum psg_conv_flags
# enum psg_flag: bits of the per-step NaN/Inf flag word
FLAG_NOISY_BAD, FLAG_T_RANGE, FLAG_PRED_BAD, FLAG_LOSS_BAD, FLAG_FALLBACK, FLAG_INPUT_BAD, FLAG_SKIP_MASK = 1, 2, 4, 8, 16, 32, 47

c_void_p, c_int, c_int64, c_float, c_uint64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64


class ConvDesc(C.Structure):
    """struct psg_conv_desc (include/psg_hip.h)."""
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Cin", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("transposed", C.c_int32), ("act", C.c_int32),
        ("alpha", C.c_float), ("drop_p", C.c_float), ("flags", C.c_int32), ("drop_seed", C.c_uint64),
        ("ldx", C.c_int64), ("ldy", C.c_int64), ("ld_rowadd", C.c_int64), ("ld_residual", C.c_int64),
        ("ld_preact", C.c_int64), ("ld_dact", C.c_int64), ("ldw", C.c_int64),
        ("x", c_void_p), ("w", c_void_p), ("y", c_void_p), ("bias", c_void_p), ("rowadd", c_void_p),
        ("residual", c_void_p), ("preact", c_void_p), ("dact_u", c_void_p),
        ("ws", c_void_p), ("ws_bytes", C.c_int64),
    ]


class WgradDesc(C.Structure):
    """struct psg_wgrad_desc (include/psg_hip.h)."""
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Cin", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("accumulate", C.c_int32),
        ("dw_layout", C.c_int32), ("accumulate_bias", C.c_int32), ("scale", C.c_float), ("reserved1", C.c_int32),
        ("ldx", C.c_int64), ("lddy", C.c_int64),
        ("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("dbias", c_void_p), ("ws", c_void_p), ("ws_bytes", C.c_int64),
    ]


# name -> (restype, argtypes); every symbol include/psg_hip.h declares
SIGNATURES = {
    "psg_last_error": (C.c_char_p, []),
    "psg_version": (c_int, []),
    "psg_init": (c_int, [c_int]),
    "psg_noise_add_f32": (c_int, [c_void_p] * 7 + [c_int64, c_int64, c_int, c_int, c_void_p]),
    "psg_noise_fallback_f32": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_void_p]),
    "psg_ddpm_update_f32": (c_int, [c_void_p] * 7 + [c_int64, c_void_p]),
    "psg_sampler_update_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int] + [c_float] * 4 + [c_int64, c_void_p]),
    "psg_reparam_f32": (c_int, [c_void_p] * 4 + [c_int64, c_void_p]),
    "psg_smooth_l1_f32": (c_int, [c_void_p] * 5 + [c_float, c_float, c_int64, c_void_p, c_void_p]),
    "psg_reduce_workspace_bytes": (c_int64, []),
    "psg_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "psg_nhwc_to_nchw": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "psg_text_pool": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "psg_timestep_sinusoid": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "psg_upsample_bilinear_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64] + [c_int] * 7 + [c_void_p]),
    "psg_upsample_bilinear_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64] + [c_int] * 7 + [c_void_p]),
    "psg_add": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "psg_set_seed_source": (c_int, [c_void_p]),
    "psg_sum_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "psg_groupnorm_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64] + [c_void_p] * 4 + [c_int] * 4 + [c_float, c_int, c_int, c_void_p, c_void_p]),
    "psg_groupnorm_fwd_workspace_bytes": (c_int64, [c_int, c_int]),
    "psg_groupnorm_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64] + [c_void_p] * 4 + [c_void_p, c_int64, c_void_p, c_void_p]
                          + [c_int] * 7 + [c_void_p, c_void_p]),
    "psg_groupnorm_bwd_res": (c_int, [c_void_p, c_int64, c_void_p, c_int64] + [c_void_p] * 4 + [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]
                              + [c_int] * 7 + [c_void_p, c_void_p]),
    "psg_groupnorm_bwd_workspace_bytes": (c_int64, [c_int, c_int]),
    "psg_conv_fwd": (c_int, [C.POINTER(ConvDesc), c_void_p]),
    "psg_conv_fwd_workspace_bytes": (c_int64, [C.POINTER(ConvDesc)]),
    "psg_conv_set_pw": (c_int, [c_int]),
    "psg_conv_pw_launches": (c_int64, []),
    "psg_conv_wgrad": (c_int, [C.POINTER(WgradDesc), c_void_p]),
    "psg_conv_wgrad_workspace_bytes": (c_int64, [C.POINTER(WgradDesc)]),
    "psg_prep_weight": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "psg_kpad": (c_int64, [c_int64, c_int]),
    "psg_colsum": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "psg_colsum_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "psg_epilogue_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_float, c_float,
                                 c_uint64, c_int, c_void_p]),
    "psg_dropout_apply": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_float, c_uint64, c_float, c_int, c_void_p]),
    "psg_attn_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]
                     + [c_int] * 5 + [c_float, c_float, c_uint64, c_int, c_void_p]),
    "psg_attn_path_counts": (c_int, [c_void_p, c_void_p, c_void_p]),
    "psg_attn_set_paths": (c_int, [c_int]),
    "psg_attn_bwd": (c_int, [c_void_p, c_int64] * 5 + [c_void_p, c_void_p] + [c_void_p, c_int64] * 3
                     + [c_int] * 5 + [c_float, c_float, c_uint64, c_int, c_void_p]),
    "psg_sumsq_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "psg_adamw_f32": (c_int, [c_void_p] * 4 + [c_int64] + [c_float] * 5 + [c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "psg_adamw_dev_f32": (c_int, [c_void_p] * 4 + [c_int64, c_void_p, c_void_p, c_int] + [c_float] * 4 + [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "psg_clip_scale_f32": (c_int, [c_void_p, c_int64, c_void_p, c_float, c_void_p]),
    "psg_set_available_cus": (c_int, [c_int]),
    "psg_set_reserve_rounds": (c_int, [c_int]),
    "psg_stream_create_cu_mask": (c_int, [c_int, c_void_p]),
    "psg_stream_destroy": (c_int, [c_void_p]),
    "psg_profile_begin": (c_int, []),
    "psg_profile_end": (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    "psg_profile_bytes": (c_int, [c_void_p, c_int]),
}

_lib = None
_inited = set()


class PsgError(RuntimeError):
    pass


def load():
    """dlopen libpsg_hip.so and declare every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PsgError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pokemon_sprite_generator_amd/csrc`). There is no CPU/eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().psg_last_error()
        raise PsgError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def init(device_index: int):
    lib = load()
    if device_index not in _inited:
        check(lib.psg_init(int(device_index)), "psg_init")
        _inited.add(device_index)
    return lib


AVAIL_CUS = [256, 0]                     # mirror of (psg_set_available_cus, psg_set_reserve_rounds): ops caches split-K plans per value


def set_available_cus(device_index: int, n: int, rounds: int = 0):
    """psg_set_available_cus (+ psg_set_reserve_rounds) and the host-side mirror (n = 0: the whole chip)."""
    lib = init(device_index)
    check(lib.psg_set_available_cus(int(n)), "psg_set_available_cus")
    check(lib.psg_set_reserve_rounds(int(rounds)), "psg_set_reserve_rounds")
    AVAIL_CUS[0], AVAIL_CUS[1] = (256 if n == 0 else int(n)), int(rounds)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """The current HIP stream of the current device as a C pointer (honours `with torch.cuda.stream(...)`).  The raw
    accessor skips the Stream object `torch.cuda.current_stream()` builds: 8 us a call, ~1000 calls per train step."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(dt):
    if dt == torch.float32:
        return PSG_F32
    if dt == torch.bfloat16:
        return PSG_BF16
    raise PsgError(f"unsupported compute dtype {dt}")


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ---------------------------------------------------------------------------
# grow-only device workspace (one per (device, stream); all kernels on one stream run in order, so they share scratch)
# ---------------------------------------------------------------------------
class WorkspacePool:
    """key -> buffer, grown by replacement.  A captured hipGraph holds RAW POINTERS into the buffer of its capture stream, and
    torch recycles stream handles (a pool of 32 per device), so the rules are:
      * while frozen (a capture is open) a request that would replace a buffer raises - growth belongs to the eager warm-up;
      * a graph owner `hold()`s the buffer its capture used: it keeps a reference (the memory cannot be freed under the
        graph, whatever later happens to the key) and the key is marked held - a later, larger request on a recycled handle
        then allocates a NEW buffer for the key and leaves the held one alone instead of freeing it;
      * `drop()` (owner closed) releases the hold and, when the key's current buffer is the held one, the entry - so
        repeated graph runs do not leave one buffer per recycled stream behind."""

    def __init__(self, alloc):
        self._alloc = alloc
        self._bufs = {}
        self._held = {}          # key -> list of buffers graph owners still point into
        self.frozen = False

    def get(self, key, nbytes, *alloc_args):
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            if self.frozen:
                raise PsgError(f"workspace of {nbytes} bytes requested inside a graph capture but the warm-up sized "
                               f"{0 if buf is None else buf.numel()} on this stream: run the eager warm-up on the capture stream")
            buf = self._alloc(max(int(nbytes), 1 << 20), *alloc_args)
            self._bufs[key] = buf    # (a held predecessor stays alive through self._held and its owner)
        return buf

    def hold(self, key):
        buf = self._bufs.get(key)
        if buf is None:
            raise PsgError("hold_workspace: no workspace was sized on this stream (run the eager warm-up first)")
        self._held.setdefault(key, []).append(buf)
        return key, buf

    def drop(self, handle):
        key, buf = handle
        held = self._held.get(key, [])
        for i, b in enumerate(held):
            if b is buf:
                del held[i]
                break
        if not held:
            self._held.pop(key, None)
            if self._bufs.get(key) is buf:
                del self._bufs[key]

    def __len__(self):
        return len(self._bufs)


_pool = WorkspacePool(lambda n, device: torch.empty(n, dtype=torch.uint8, device=device))


def freeze_workspaces(on: bool):
    """While frozen (a hipGraph capture is open) a request that would REPLACE a workspace raises: kernels already captured
    keep pointing at the old buffer, so growth must have happened in the eager warm-up on the same stream."""
    _pool.frozen = bool(on)


def _ws_key(device):
    di = device.index if device.index is not None else torch.cuda.current_device()
    return (di, _raw_stream(di) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream)


def workspace(nbytes: int, device):
    # one buffer per (device, stream): kernels on one stream run in order, so they can share scratch; a second
    # stream (ops: weight gradients overlapped with data gradients) gets its own
    return _pool.get(_ws_key(device), nbytes, device)


def hold_workspace(device):
    """For the owner of a captured graph, called on the capture stream after the capture: pins the buffer the captured launches
    point into (see WorkspacePool).  Returns a handle for `drop_workspace`."""
    return _pool.hold(_ws_key(device))


def drop_workspace(handle):
    if handle is not None:
        _pool.drop(handle)

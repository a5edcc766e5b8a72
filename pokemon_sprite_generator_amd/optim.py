"""Gradient arena + fused clip/AdamW (reference: improved_diffusion_trainer.py:277-283, 399-413).

ParamArena moves every parameter into ONE flat fp32 buffer (the nn.Parameters
become views, state_dict keys/shapes unchanged) and stores 3x3 conv weights in
OHWI memory order (torch channels_last): that is the order the weight-gradient
kernel produces and the prepared forward weight wants, so no permute pass runs
in either direction.  GradArena keeps every gradient in a second flat buffer
with the same offsets and strides (p.grad are persistent views) and registers
each view as a GradSink: the wgrad / column-sum / GroupNorm-backward kernels
write gradients straight into it.  The global L2 norm is then one deterministic
reduction (psg_sumsq_f32) instead of 478 `.item()` syncs, clip + AdamW is ONE
launch over the flat buffers (psg_adamw_dev_f32) and the data-parallel all-reduce
runs on large flat slices.

Ownership: a parameter belongs to at most one ParamArena and one GradArena at a
time.  Building a second arena over a parameter DISPLACES the first one: the
older arena (and the optimizer on it) raises from then on instead of silently
training on stale views.  `release()` gives the registrations back explicitly.
"""
import weakref

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .ops import GradSink, ParamShadow, SideStream, WeightCache


def _arena_offsets(params, layout=None):
    """offsets[i] of parameter i in the flat buffer.  `layout` (a permutation of the parameter indices) is the MEMORY order -
    by default the parameter order; a model may ask for another one (UNet.arena_layout puts the 17 time_proj / text_proj
    weights next to each other so that they are ONE [sum Cout, K] operand: ops.ProjGroup)."""
    order = list(range(len(params))) if layout is None else list(layout)
    if sorted(order) != list(range(len(params))):
        raise ValueError("arena layout must be a permutation of the parameter indices")
    offsets, off = [0] * len(params), 0
    for i in order:
        offsets[i] = off
        off += (params[i].numel() + 7) // 8 * 8      # 32-byte steps: fp32 views and their bf16 shadows stay 16-byte aligned
    return offsets, off


class ArenaDisplaced(_lib.PsgError):
    pass


class ParamArena:
    """Flat fp32 master copy of the parameters; `p.data` become views of it (in-place loads such as
    load_state_dict keep the binding).  4-D weights with a spatial kernel are stored OHWI (channels_last)."""

    _owner = {}          # id(param) -> weakref(ParamArena) that currently backs the parameter's storage

    def __init__(self, params, layout=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("ParamArena: no trainable parameters")
        dev = self.params[0].device
        self.offsets, self.numel = _arena_offsets(self.params, layout)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self._displaced_by = None
        me = weakref.ref(self)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                if p.dtype != torch.float32:
                    raise TypeError("ParamArena: master parameters must be fp32")
                old = ParamArena._owner.get(id(p))
                old = old() if old is not None else None
                if old is not None and old is not self and any(q is p for q in old.params):
                    old._displace(self)
                if p.dim() == 4 and p.shape[2] * p.shape[3] > 1:
                    O, I, kh, kw = p.shape
                    v = self.flat[o:o + p.numel()].view(O, kh, kw, I).permute(0, 3, 1, 2)
                else:
                    v = self.flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                ParamArena._owner[id(p)] = me
        WeightCache.drop(self.params)    # their storage moved: release the prepared copies (other models keep theirs)
        self.shadow = None               # bf16 copy of `flat`, kept current by FusedAdamW (enable_shadow)
        self._shadow_epoch = -1
        self._versions = None

    def _displace(self, by):
        self._displaced_by = by
        ParamShadow.unregister(self)
        self.shadow = None               # (10+ GB at full width: release it now)

    def check_alive(self):
        if self._displaced_by is not None:
            raise ArenaDisplaced("this ParamArena no longer backs its parameters: a newer arena (a second stepper over the same "
                                 "model) re-bound them; build one stepper per model, or step only the newest")

    def release(self):
        """Give up the shadow registrations (the parameters keep pointing into this arena's buffer)."""
        ParamShadow.unregister(self)
        self.shadow = None

    def enable_shadow(self):
        """Keep a bf16 shadow of the flat masters: FusedAdamW writes it in its update pass, ops.WeightCache uses
        its slices as prepared forward weights.  A parameter changed by anything else (load_state_dict, manual
        copy_) is detected through its torch version counter and prepared from the fp32 master until the next step."""
        self.check_alive()
        if self.shadow is None:
            self.shadow = self.flat.to(torch.bfloat16)
            for i, p in enumerate(self.params):
                ParamShadow.register(p, self, i)
        else:
            self.shadow.copy_(self.flat)
        self.mark_shadow_current()

    def mark_shadow_current(self):
        self._shadow_epoch = WeightCache.epoch
        self._versions = [p._version for p in self.params]

    def shadow_slice(self, index):
        if self.shadow is None or self._shadow_epoch != WeightCache.epoch:
            return None
        if isinstance(index, tuple):                 # (member indices of a contiguous run): the shadow of the whole run
            if any(self.params[i]._version != self._versions[i] for i in index):
                return None
            o0, last = self.offsets[index[0]], index[-1]
            return self.shadow[o0:self.offsets[last] + self.params[last].numel()]
        p = self.params[index]
        if p._version != self._versions[index]:
            return None
        o = self.offsets[index]
        return self.shadow[o:o + p.numel()]

    def like(self, flat, index):
        """View of another flat buffer with parameter `index`'s offset, shape and strides."""
        p = self.params[index]
        return flat.as_strided(p.shape, p.stride(), self.offsets[index])

    def broadcast(self, src=0, group=None):
        """Data-parallel start-up: every rank takes rank `src`'s masters (one collective over the flat buffer), so
        replicas are equal by construction instead of by identical RNG seeding."""
        torch.distributed.broadcast(self.flat, src=src, group=group)
        if self.shadow is not None:
            self.shadow.copy_(self.flat)
        WeightCache.invalidate()
        if self.shadow is not None:
            self.mark_shadow_current()


class GradArena:
    def __init__(self, params, on_ready=None, layout=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradArena: no trainable parameters")
        dev = self.params[0].device
        self.offsets, self.numel = _arena_offsets(self.params, layout)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.normsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.on_ready = on_ready
        self.views = []
        self.entries = []
        self.extra_entries = []            # sinks of virtual parameters over runs of this arena (ops.ProjGroup): reset with ours
        self._displaced = False
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            v = self.flat.as_strided(p.shape, p.stride(), o)      # the parameter's own memory order
            self.views.append(v)
            p.grad = v
            self.entries.append(GradSink.register(p, v, i, self._ready, owner=self))

    # GradSink calls this when another arena registers one of our parameters
    def displaced(self, param):
        if not self._displaced:
            self._displaced = True
            GradSink.unregister(self)

    def check_alive(self):
        if self._displaced:
            raise ArenaDisplaced("this GradArena no longer receives its parameters' gradients: a newer arena (a second stepper over "
                                 "the same model) registered them; build one stepper per model, or step only the newest")
        for e in self.entries[:1] + self.entries[-1:]:             # cheap sanity: our sinks are still the registered ones
            if GradSink.get(e.param) is not e:
                raise ArenaDisplaced("GradArena: gradient sink registrations were cleared behind this arena's back")

    def release(self):
        GradSink.unregister(self)
        self._displaced = True

    def _ready(self, index):
        if self.on_ready is not None:
            self.on_ready(index)

    def zero(self):
        """Start of a step: nothing is memset — every sink is overwritten by its first gradient kernel."""
        self.check_alive()
        for e in self.entries:
            e.written = False
        for e in self.extra_entries:
            e.written = False
        for p, v in zip(self.params, self.views):       # re-attach if someone set grads to None
            if p.grad is not v:
                p.grad = v

    def finalize(self):
        """After backward: join the weight-gradient side stream; parameters that received no gradient this step get
        an explicit zero."""
        self.check_alive()
        SideStream.join(self.flat.device)
        for e in self.entries:
            if not e.written:
                e.view.zero_()
                GradSink.done(e)

    def grad_norm_sq(self):
        """Device scalar sum(g^2) over all parameters (padding is zero)."""
        SideStream.join(self.flat.device)
        lib = _lib.init(self.flat.device.index)
        ws = _lib.workspace(lib.psg_reduce_workspace_bytes(), self.flat.device)
        check(lib.psg_sumsq_f32(ptr(self.flat), self.numel, ptr(self.normsq), 0, ptr(ws), stream_ptr()), "psg_sumsq_f32")
        return self.normsq


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt-ratio)
    with gradient clipping folded in.  state_dict layout matches torch.optim.AdamW
    ('step', 'exp_avg', 'exp_avg_sq' per parameter), so reference checkpoints interchange.

    With `param_arena` + `grad_arena` (same parameters, same order) the whole update is one launch over
    the flat buffers; the per-parameter state tensors are views of two more flat buffers.  In that mode the step
    count lives ON THE DEVICE (`step_dev`): the kernel advances it only when the step was not skipped by the NaN
    flag, and reads the learning rate from a device table indexed by it (`set_lr_table`: entry k = the lr after k
    scheduler steps) - the reference's `continue` on a bad batch (no optimizer.step, no scheduler.step,
    improved_diffusion_trainer.py:353-393) without a host read per step.  `steps_done()` reads the counter (one sync)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, param_arena=None, grad_arena=None):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._pa, self._ga = None, None
        if param_arena is not None and grad_arena is not None:
            same = (len(params) == len(param_arena.params) == len(grad_arena.params)
                    and all(a is b and a is c for a, b, c in zip(params, param_arena.params, grad_arena.params))
                    and param_arena.offsets == grad_arena.offsets)
            if not same:
                raise ValueError("FusedAdamW: arenas do not cover the optimizer's parameters in order")
            self._pa, self._ga = param_arena, grad_arena
            dev = param_arena.flat.device
            self._m = torch.zeros_like(param_arena.flat)
            self._v = torch.zeros_like(param_arena.flat)
            self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            self._lr_table, self._b1_table, self._lr_host, self._lr_sched = None, None, None, False
            for i, p in enumerate(params):
                self.state[p] = {"step": torch.tensor(0.0), "exp_avg": param_arena.like(self._m, i),
                                 "exp_avg_sq": param_arena.like(self._v, i)}

    # ---- device-side step count / schedule ------------------------------------------------------------------
    def steps_done(self) -> int:
        """Optimizer steps that really happened (skipped NaN batches excluded).  Host sync."""
        if self._pa is None:
            steps = {int(st["step"]) for st in self.state.values() if "step" in st}
            return max(steps) if steps else 0
        return int(self.step_dev.item())

    def set_lr_table(self, values, beta1_values=None):
        """Learning rate (and optionally beta1: OneCycleLR cycles it) per completed-step count: values[k] is used for
        optimizer step k+1 (the last entry repeats)."""
        t = torch.as_tensor(list(values) if not torch.is_tensor(values) else values, dtype=torch.float32)
        if t.numel() < 1:
            raise ValueError("set_lr_table: empty table")
        self._lr_table = t.to(self._pa.flat.device).contiguous()
        self._b1_table = None
        if beta1_values is not None:
            b = torch.as_tensor(list(beta1_values) if not torch.is_tensor(beta1_values) else beta1_values, dtype=torch.float32)
            if b.numel() != t.numel():
                raise ValueError("set_lr_table: lr and beta1 tables differ in length")
            self._b1_table = b.to(self._pa.flat.device).contiguous()
        self._lr_sched = True

    def sched_at(self, k: int):
        """(lr, beta1) the update after k completed steps uses."""
        b1 = float(self.param_groups[0]["betas"][0])
        if self._lr_table is None or not self._lr_sched:
            return float(self.param_groups[0]["lr"]), b1
        i = min(int(k), self._lr_table.numel() - 1)
        return float(self._lr_table[i].item()), (float(self._b1_table[i].item()) if self._b1_table is not None else b1)

    def _lr_ptr(self):
        if not self._lr_sched:                       # constant lr from the param group (re-uploaded when the host changes it)
            lr = float(self.param_groups[0]["lr"])
            if self._lr_table is None or self._lr_host != lr:
                self._lr_table = torch.full((1,), lr, dtype=torch.float32, device=self._pa.flat.device)
                self._lr_host = lr
        return self._lr_table

    def _sync_steps(self):
        if self._pa is not None:
            n = float(self.steps_done())
            for st in self.state.values():
                st["step"].fill_(n)

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._pa is None:
            return
        steps = set()
        for i, p in enumerate(self._pa.params):           # re-bind the loaded moments to the flat buffers
            st = self.state[p]
            m, v = self._pa.like(self._m, i), self._pa.like(self._v, i)
            m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"] = m, v
            st["step"] = torch.tensor(float(st["step"]))
            steps.add(int(st["step"]))
        if len(steps) != 1:
            raise ValueError(f"FusedAdamW: flat update needs one common step count, checkpoint has {sorted(steps)}")
        self.step_dev.fill_(steps.pop())

    @torch.no_grad()
    def step(self, normsq=None, max_norm=0.0, skip_flag=None, grad_scale=1.0):
        nptr = ptr(normsq) if normsq is not None else None
        fptr = ptr(skip_flag) if skip_flag is not None else None
        if self._pa is not None:
            self._pa.check_alive()
            self._ga.check_alive()
            group = self.param_groups[0]
            b1, b2 = group["betas"]
            lib = _lib.init(self._pa.flat.device.index)
            shadow = self._pa.shadow
            table = self._lr_ptr()
            check(lib.psg_adamw_dev_f32(ptr(self._pa.flat), ptr(self._ga.flat), ptr(self._m), ptr(self._v), self._pa.numel,
                                        ptr(table), ptr(self._b1_table) if self._lr_sched else None, table.numel(), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                        ptr(self.step_dev), nptr, float(max_norm), fptr, ptr(shadow), stream_ptr()), "psg_adamw_dev_f32")
            WeightCache.invalidate()
            if shadow is not None:
                self._pa.mark_shadow_current()       # the kernel just rewrote it (a skipped NaN step leaves both untouched)
            return
        # generic path (parameters not in an arena): host-side step counts, so the skip decision needs a host read
        if skip_flag is not None and (int(skip_flag.item()) & _lib.FLAG_SKIP_MASK):
            return
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
                g = p.grad
                if g.stride() != p.stride():               # the kernel walks raw memory: same order required
                    g = torch.empty_like(p).copy_(g)
                check(lib.psg_adamw_f32(ptr(p), ptr(g), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), p.numel(), float(group["lr"]),
                                        float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), int(st["step"]),
                                        nptr, float(max_norm), None, None, stream_ptr()), "psg_adamw_f32")
        WeightCache.invalidate()     # parameters changed through raw pointers

"""ImprovedDiffusionTrainer mirror (reference: src/training/improved_diffusion_trainer.py:77-692).

Keeps the reference's surface — `__init__(config, vae_checkpoint_path, experiment_name)`,
`train()`, `train_epoch`, `validate_epoch`, `ddpm_sample`, `generate_samples`,
`save_checkpoint`, `load_checkpoint` — so `train_3stage.py:130-140` runs unchanged
when this class is imported in place of the reference's, and adds the facade
the north-star names: `train_step(latents, text_emb, t, noise=None)` (the loop
body :363-413) and `sample(text_emb, num_samples, fast_sampling, noise_fn=None)`
(:508-569).

The frozen VAE either side of the path is this package's own (`vae.VAEEncoder` /
`vae.VAEDecoder`, same kernels).  Out-of-scope collaborators (BERT text encoder,
data loaders, TensorBoard) are NOT re-implemented: they are taken from the
reference package (`src.models`, `src.data`) when it is importable, or injected
through `components=` (tests, benchmarks with synthetic latents).

Bad-batch semantics are the reference's (:353-393): a batch whose text embedding,
latent, noisy latent, prediction or loss is non-finite is skipped — no optimizer
step, no scheduler step, no `global_step` increment, not part of the epoch mean —
and a noisy latent rescued by add_noise's `x0 + 0.1*noise` fallback (:61-63) trains.
The decision is made ON THE DEVICE (flag word, `psg_flag` in include/psg_hip.h):
the step counter and the lr / beta1 schedule live in device memory and advance only
on steps that happened, so the loop never reads the flag on the host.
"""
import logging
import os
from pathlib import Path
from typing import Any, Dict, Optional

import torch

from . import _lib
from . import ops as ops_mod
from ._lib import FLAG_INPUT_BAD, FLAG_LOSS_BAD, FLAG_NOISY_BAD, FLAG_PRED_BAD, FLAG_SKIP_MASK, FLAG_T_RANGE, check, ptr, stream_ptr
from .ddp import BucketedAllReduce, ShardedLoader, dist_info, rank_generator
from .optim import FusedAdamW, GradArena, ParamArena
from .scheduler import NoiseScheduler
from .unet import UNet


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass

    def close(self):
        pass


class DiffusionStepper:
    """The accelerated hot path: add_noise -> U-Net -> SmoothL1 -> backward -> [all-reduce] -> clip -> AdamW.

    Everything stays on the device; the only host read is the caller's choice
    (`.item()` on the returned tensors).  The reference's 478 per-parameter
    `.item()` syncs (:399-404) and 5 full-tensor NaN scans (:353-393) are
    replaced by one fused norm reduction and a device-side flag word.
    """

    def __init__(self, unet: UNet, noise_scheduler: NoiseScheduler, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01,
                 eps=1e-6, max_grad_norm=1.0, optimizer_type="adamw", distributed=None, bucket_bytes=64 << 20,
                 grad_bucket_dtype=torch.float32, ddp_cu_reserve=None):
        self.unet, self.noise_scheduler = unet, noise_scheduler
        self.max_grad_norm = max_grad_norm
        self.device = next(unet.parameters()).device
        if self.device.type != "cuda":
            raise _lib.PsgError("DiffusionStepper needs the U-Net on a GPU (HIP path only)")
        self.lib = _lib.init(self.device.index if self.device.index is not None else torch.cuda.current_device())
        self.reducer = None
        # distributed="force": run the gradient exchange and the flag reduce even in a process group of ONE rank (the RCCL
        # path on a 1-GPU box; the step's results are those of the non-distributed step bit for bit)
        force_single = distributed == "force"
        if force_single and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            raise _lib.PsgError('DiffusionStepper(distributed="force") needs an initialised process group')
        if distributed is None:
            distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        distributed = bool(distributed)
        # memory order of the two arenas: the model may ask for one (UNet.arena_layout: the time / text projections adjacent)
        layout = unet.arena_layout([p for p in unet.parameters() if p.requires_grad]) if hasattr(unet, "arena_layout") else None
        self.params = ParamArena(unet.parameters(), layout=layout)        # flat fp32 masters, conv weights OHWI
        if distributed:
            self.params.broadcast(src=0)                    # replicas equal by construction, not by RNG seeding
        if getattr(unet, "compute_dtype", None) == torch.bfloat16 and optimizer_type == "adamw" and not os.environ.get("PSG_NO_SHADOW"):
            self.params.enable_shadow()                     # AdamW also emits next step's bf16 forward weights
        self.arena = GradArena(unet.parameters(), on_ready=lambda i: self.reducer.on_ready(i) if self.reducer is not None else None,
                               layout=layout)
        # Adam (non-decoupled decay) is torch's when asked for (:285-291); AdamW is the fused kernel (:277-283)
        if optimizer_type == "adamw":
            self.optimizer = FusedAdamW(self.arena.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                        param_arena=self.params, grad_arena=self.arena)
        else:
            self.optimizer = torch.optim.Adam(self.arena.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.reducer = BucketedAllReduce(self.arena.flat, self.arena.params, self.arena.offsets, bucket_bytes,
                                         bucket_dtype=grad_bucket_dtype, force_single=force_single,
                                         cu_reserve=ddp_cu_reserve) if distributed else None
        if hasattr(unet, "bind_proj_group"):
            unet.bind_proj_group(self.params, self.arena)       # (after enable_shadow: the group's bf16 operand is a run of it)
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)      # bits: enum psg_flag (include/psg_hip.h)
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.criterion_beta = 0.1                                               # nn.SmoothL1Loss(beta=0.1) :300
        self.host_steps = 0                                                     # (torch.optim.Adam branch only)
        # data parallel: this rank's own stream for t / noise / VAE eps (None = torch's default generator); see ddp.rank_generator
        self.generator = rank_generator(self.device) if distributed else None
        self._graphs = []                          # live GraphedTrainStep objects of this stepper (closed with it)

    def close(self):
        """Release the process-wide registrations (gradient sinks, bf16 shadow, the graph-replay seed word) of this stepper."""
        for g in list(self._graphs):               # (each holds one count of the seed source and its capture stream's workspace)
            g.close()
        if hasattr(self.unet, "bind_proj_group") and getattr(self.unet, "_proj_group", None) is not None and self.unet._proj_group.ga is self.arena:
            self.unet.bind_proj_group(None, None)
        self.arena.release()
        self.params.release()

    def randn_like(self, x):
        """randn_like from this rank's stream (:373)."""
        if self.generator is None:
            return torch.randn_like(x)
        return torch.randn(x.shape, dtype=x.dtype, device=x.device, generator=self.generator)

    def randint_t(self, n):
        """Timesteps of a batch (:366) from this rank's stream."""
        T = self.noise_scheduler.num_timesteps
        if self.generator is None:
            return torch.randint(0, T, (n,), device=self.device)
        return torch.randint(0, T, (n,), device=self.device, generator=self.generator)

    # ---- fused SmoothL1 + dL/d(eps_hat) ------------------------------------------------------------
    def smooth_l1(self, pred, target, want_grad=True):
        pred_c, tgt = pred.detach().contiguous(), target.detach().contiguous().float()
        grad = torch.empty_like(pred_c) if want_grad else None
        ws = _lib.workspace(self.lib.psg_reduce_workspace_bytes(), self.device)
        check(self.lib.psg_smooth_l1_f32(ptr(pred_c), ptr(tgt), ptr(grad), ptr(self.loss), ptr(self.flag), self.criterion_beta, 1.0,
                                         pred_c.numel(), ptr(ws), stream_ptr()), "psg_smooth_l1_f32")
        return self.loss, grad

    def _reduce_flag(self):
        """All ranks must skip together: if ANY rank has a skip bit, every rank ends up with a non-zero skip field
        (RCCL has no bitwise-OR reduction: MAX of the masked field keeps 'non-zero'; the local informational bits stay)."""
        skip = self.flag & FLAG_SKIP_MASK
        torch.distributed.all_reduce(skip, op=torch.distributed.ReduceOp.MAX)
        self.flag.copy_((self.flag & ~FLAG_SKIP_MASK) | skip)

    def train_step(self, latents, text_emb, t, noise=None, lr=None, pre_flag=None) -> Dict[str, torch.Tensor]:
        """One optimizer step on a batch of clean latents (the body of train_epoch, :363-413).

        latents [B,8,27,27] fp32 (un-clamped VAE output), text_emb [B,S,text_dim], t [B] int64, optional
        noise (default randn_like).  `pre_flag` (int32 device tensor or None) is OR-ed into the step's flag word
        (the caller's own input checks, :353,359).  Returns device tensors {'loss','grad_norm','nan_flag'} (no sync);
        a step whose flag has a bit of FLAG_SKIP_MASK leaves parameters, optimizer state, step count and schedule
        untouched."""
        if not self.unet.training:             # (module.train() walks all ~390 submodules: 1.6 ms of host time per call)
            self.unet.train()
        if ops_mod.SeedSource.enabled():       # fixed per-site seeds + a device word that changes per step (graph replay)
            from .unet import _SeedStream
            _SeedStream.counter = 0
        if noise is None:
            noise = self.randn_like(latents)
        self.flag.zero_()
        if pre_flag is not None:
            self.flag |= pre_flag.to(device=self.device, dtype=torch.int32).reshape(-1)[:1]
        noisy = self.noise_scheduler.add_noise(latents, noise, t, clamp=True, flag=self.flag)     # :363, :374
        self.arena.zero()                                                                           # :380
        eps_hat = self.unet(noisy, t, text_emb)                                                     # :381
        loss, dpred = self.smooth_l1(eps_hat, noise)                                                # :388
        eps_hat.backward(dpred)                                                                     # :396
        self.arena.finalize()                      # (joins the weight-gradient side stream)
        if self.reducer is not None:
            self.reducer.finish()
            self._reduce_flag()
        normsq = self.arena.grad_norm_sq()                                                          # :399-404
        if lr is not None:
            for g in self.optimizer.param_groups:
                g["lr"] = lr
        if isinstance(self.optimizer, FusedAdamW):
            self.optimizer.step(normsq=normsq, max_norm=self.max_grad_norm, skip_flag=self.flag)    # :410-412
        elif (int(self.flag.item()) & FLAG_SKIP_MASK) == 0:      # torch.optim.Adam: the host decides (one sync per step)
            check(self.lib.psg_clip_scale_f32(ptr(self.arena.flat), self.arena.numel, ptr(normsq), float(self.max_grad_norm),
                                              stream_ptr()), "psg_clip_scale_f32")
            self.optimizer.step()
            self.host_steps += 1
            from .ops import WeightCache
            WeightCache.invalidate()
        ops_mod.SeedSource.advance()
        return {"loss": loss.clone(), "grad_norm": normsq.sqrt(), "nan_flag": self.flag.clone()}

    def capture_train_step(self, latents, text_emb, t, warmup=2):
        """hipGraph of one whole train step (add_noise .. AdamW, ~1200 launches) for fixed shapes: returns a `GraphedTrainStep`
        whose `run(latents, text_emb, t)` copies the batch into the captured buffers and replays.  The step has no host
        reads (flags, step count and schedule live on the device) and its dropout masks keep changing between replays
        through `ops.SeedSource`; noise is drawn inside the graph (torch's graph-safe generator).  `warmup` eager steps run
        first ON THE GIVEN BATCH (they train); capture itself executes nothing.  Single process only (no gradient
        all-reduce inside the graph).  23 ms of launches per step become one replay: the step is GPU-bound at any batch.
        MODE SWITCH: capturing enables `ops.SeedSource` (fixed per-site dropout seeds + a device word advanced every step)
        for the whole process - eager steps of this and any other stepper draw their masks that way too from then on -
        until the LAST live graph is closed (`GraphedTrainStep.close()`, or this stepper's `close()`; the source is reference
        counted, one count per graph: the device word is baked into captured kernel arguments)."""
        if self.reducer is not None:
            raise NotImplementedError("capture_train_step: data-parallel steps are not captured")
        if not isinstance(self.optimizer, FusedAdamW):
            raise NotImplementedError("capture_train_step needs the fused optimizer (device-side step count)")
        return GraphedTrainStep(self, latents, text_emb, t, warmup)

    def autotune_exchange(self, latents, text_emb, t, trials=3):
        """Data parallel: measure which gradient-exchange mode is fastest on THIS machine (ddp.BucketedAllReduce.autotune: real
        train steps on the given batch, every rank takes part) WITHOUT training - masters, moments, step count and the bf16
        shadow are snapshotted (10 GB at full width) and put back, so the model, the optimizer and the schedule are exactly
        where they were; only the random streams have moved on.  Returns the record (None in a single process)."""
        if self.reducer is None or not self.reducer.active or not isinstance(self.optimizer, FusedAdamW):
            return None
        opt = self.optimizer
        snap = (self.params.flat.clone(), opt._m.clone(), opt._v.clone(), opt.step_dev.clone(),
                None if self.params.shadow is None else self.params.shadow.clone())
        try:
            rec = self.reducer.autotune(lambda: self.train_step(latents, text_emb, t), trials=trials)
        finally:
            with torch.no_grad():
                self.params.flat.copy_(snap[0]); opt._m.copy_(snap[1]); opt._v.copy_(snap[2]); opt.step_dev.copy_(snap[3])
                if snap[4] is not None:
                    self.params.shadow.copy_(snap[4])
            ops_mod.WeightCache.invalidate()
            if self.params.shadow is not None:
                self.params.mark_shadow_current()
        return rec

    def steps_done(self) -> int:
        """Optimizer steps that really happened (host sync for the fused optimizer)."""
        return self.optimizer.steps_done() if isinstance(self.optimizer, FusedAdamW) else self.host_steps

    @torch.no_grad()
    def eval_loss(self, latents, text_emb, t, noise=None):
        """validate_epoch body (:465-486)."""
        self.unet.eval()
        if noise is None:
            noise = self.randn_like(latents)
        self.flag.zero_()
        noisy = self.noise_scheduler.add_noise(latents, noise, t, clamp=True, flag=self.flag)
        eps_hat = self.unet(noisy, t, text_emb)
        loss, _ = self.smooth_l1(eps_hat, noise, want_grad=False)
        return loss.clone(), self.flag.clone()

    @staticmethod
    def _sample_set(t, i, tv, t_dev, z, rnd, x):
        """Per-step inputs, written in place (the captured graph reads these buffers)."""
        tv.fill_(t)
        t_dev.fill_(t)
        if t > 0:
            z.copy_(rnd(i, tuple(x.shape)).to(device=x.device, dtype=torch.float32))

    @torch.no_grad()
    def sample(self, text_emb, num_samples, fast_sampling=True, noise_fn=None, latent_dim=8, hw=27, trace=None, use_graph=None,
               max_steps=None):
        """ddpm_sample (:508-569): x <- (x - c2*eps)/sqrt(alpha_t) [+ sqrt(beta_t)*z if t>0], t strided by 50 when fast.
        noise_fn(i, shape) -> tensor supplies x_T (i = -1) and the per-step z (tests inject it); default torch.randn.
        `max_steps` runs only the first that many steps of the chain (benchmarks / property tests of the 1000-step loop).
        See SamplerRun for the hipGraph form (use_graph=True or PSG_GRAPH=1; bit-identical to the eager loop)."""
        run = SamplerRun(self, text_emb, num_samples, fast_sampling, noise_fn, latent_dim, hw, use_graph, max_steps)
        while run.remaining():
            run.step()
            if trace is not None:
                trace.append(run.x.clone())
        return run.x

    def sampler(self, text_emb, num_samples, fast_sampling=True, noise_fn=None, latent_dim=8, hw=27, use_graph=None, max_steps=None):
        """The same chain, one `step()` at a time (bench.py times single denoising steps)."""
        return SamplerRun(self, text_emb, num_samples, fast_sampling, noise_fn, latent_dim, hw, use_graph, max_steps)


class GraphedTrainStep:
    """See DiffusionStepper.capture_train_step."""

    def __init__(self, stepper, latents, text_emb, t, warmup):
        from .ops import SeedSource, WeightCache
        self.stepper = stepper
        dev = stepper.device
        SeedSource.acquire(dev)
        self._ws_hold = None
        self._closed = False
        self.lat, self.txt, self.t = latents.detach().clone(), text_emb.detach().clone(), t.detach().clone()
        # Warm-up AND capture run on ONE stream of ours: the scratch workspace is keyed by (device, stream), so the eager
        # steps size exactly the buffers the captured launches will use and nothing is (re)allocated inside the capture.
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                stepper.train_step(self.lat, self.txt, self.t)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            _lib.freeze_workspaces(True)           # a growing request inside the capture raises instead of replacing a captured buffer
            try:
                with torch.cuda.graph(self.graph, stream=side):
                    self.out = stepper.train_step(self.lat, self.txt, self.t)
            finally:
                _lib.freeze_workspaces(False)
            # the captured launches hold raw pointers into this stream's scratch buffer: pin it (torch recycles stream handles)
            self._ws_hold = _lib.hold_workspace(dev)
        self._stream = side
        torch.cuda.current_stream(dev).wait_stream(side)
        WeightCache.invalidate()
        stepper._graphs.append(self)

    def close(self):
        """Release what the graph pinned: its capture stream's workspace and one count of the dropout seed source."""
        if self._closed:
            return
        self._closed = True
        from .ops import SeedSource
        self.graph = None
        _lib.drop_workspace(self._ws_hold)
        self._ws_hold = None
        SeedSource.release()
        if self in self.stepper._graphs:
            self.stepper._graphs.remove(self)

    def run(self, latents=None, text_emb=None, t=None):
        from .ops import WeightCache
        if self._closed:
            raise _lib.PsgError("GraphedTrainStep.run after close()")
        if latents is not None:
            self.lat.copy_(latents)
        if text_emb is not None:
            self.txt.copy_(text_emb)
        if t is not None:
            self.t.copy_(t)
        self.graph.replay()
        WeightCache.invalidate()          # the replay moved the weights: entries prepared by eager code are stale
        return self.out


class SamplerRun:
    """One ddpm_sample chain (:508-569), advanced step by step.

    use_graph=True (or PSG_GRAPH=1) captures the loop body (U-Net forward + update, ~700 launches) ONCE into a hipGraph
    (torch.cuda.CUDAGraph over the library's launches on the capture stream) and replays it per step: the timestep lives
    in device memory (the embedding and the update kernel both read it there) and z is refilled in place, so the
    captured work is identical for every step; results are bit-identical to the eager path (tests/test_unet_gpu.py).
    Step 0 always runs eagerly - it sizes the workspace and fills the prepared-weight cache (no allocation and no weight
    preparation may happen inside the capture) - and the capture happens at step 1.  A capture failure falls back to the
    eager loop with a warning."""

    @torch.no_grad()
    def __init__(self, stepper, text_emb, num_samples, fast_sampling=True, noise_fn=None, latent_dim=8, hw=27, use_graph=None,
                 max_steps=None):
        self.st = stepper
        stepper.unet.eval()
        dev = stepper.device
        self.rnd = noise_fn if noise_fn is not None else (lambda i, shape: torch.randn(shape, device=dev))
        # (own copy: the chain updates x in place and must not write into the caller's x_T)
        self.x = self.rnd(-1, (num_samples, latent_dim, hw, hw)).to(device=dev, dtype=torch.float32).contiguous().clone()
        sch = stepper.noise_scheduler.to(dev)
        self.c1, self.c2, self.sg = sch.step_tables(dev)
        T = sch.num_timesteps
        steps = list(range(0, T, 50)) if fast_sampling else list(range(T))
        self.order = list(reversed(steps))
        if max_steps is not None:
            self.order = self.order[:max_steps]
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.tv = torch.zeros((num_samples,), device=dev, dtype=torch.long)
        self.z = torch.zeros_like(self.x)
        self.text = text_emb.to(dev)
        if use_graph is None:
            use_graph = bool(os.environ.get("PSG_GRAPH"))
        self.use_graph = bool(use_graph) and len(self.order) > 2
        self.graph = None
        self.i = 0
        # graph mode: eager step 0, the capture and every replay run on ONE stream of ours - the scratch workspace is keyed
        # by (device, stream), so step 0 sizes exactly the buffers the captured launches use (nothing is allocated or
        # replaced inside the capture: _lib.freeze_workspaces enforces it)
        self._stream = torch.cuda.Stream(device=dev) if self.use_graph else None
        self._ws_hold = None

    def close(self):
        """Drop the graph and the scratch buffer it pinned (also runs when the chain completes and on garbage collection)."""
        self.graph = None
        if self._ws_hold is not None:
            _lib.drop_workspace(self._ws_hold)
            self._ws_hold = None

    def __del__(self):
        try:
            self.close()
        except Exception:                            # noqa: BLE001 - interpreter shutdown
            pass

    def remaining(self):
        return len(self.order) - self.i

    def _body(self):
        st = self.st
        eps = st.unet(self.x, self.tv, self.text).contiguous()
        check(st.lib.psg_ddpm_update_f32(ptr(self.x), ptr(eps), ptr(self.z), ptr(self.c1), ptr(self.c2), ptr(self.sg), ptr(self.t_dev),
                                         self.x.numel(), stream_ptr()), "psg_ddpm_update_f32")

    @torch.no_grad()
    def step(self):
        if self.i >= len(self.order):
            raise StopIteration("sampler chain is complete")
        self.st.unet.eval()
        DiffusionStepper._sample_set(self.order[self.i], self.i, self.tv, self.t_dev, self.z, self.rnd, self.x)
        if self._stream is None:
            self._body()
        else:
            cur = torch.cuda.current_stream(self.st.device)
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                self._graph_step()
            cur.wait_stream(self._stream)
        self.i += 1
        if self.i >= len(self.order):
            self.close()
        return self.x

    def _graph_step(self):
        if self.use_graph and self.i == 1 and self.graph is None:
            try:
                torch.cuda.synchronize(self.st.device)
                graph = torch.cuda.CUDAGraph()
                x_before = self.x.clone()
                _lib.freeze_workspaces(True)
                try:
                    with torch.cuda.graph(graph, stream=self._stream):
                        self._body()
                finally:
                    _lib.freeze_workspaces(False)
                self.x.copy_(x_before)                  # capture does not execute: the replay below runs this step
                self.graph = graph
                self._ws_hold = _lib.hold_workspace(self.st.device)     # (we are on the capture stream: its scratch is now pinned)
            except Exception as e:                      # noqa: BLE001
                logging.getLogger(__name__).warning(f"hipGraph capture of the sampler step failed ({e!r}); running eagerly")
                self.use_graph = False
        if self.graph is not None:
            self.graph.replay()
        else:
            self._body()


def schedule_tables(make_scheduler, lr, betas, total):
    """(lr[k], beta1[k]) for k = 0..total-1 completed steps, produced by running the reference's own scheduler class on a
    one-parameter dummy optimizer (OneCycleLR also cycles Adam's beta1, torch/optim/lr_scheduler.py) - no re-derivation."""
    dummy = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=lr, betas=betas)
    sch = make_scheduler(dummy)
    lrs, b1s = [], []
    for k in range(max(1, total)):
        g = dummy.param_groups[0]
        lrs.append(float(g["lr"])); b1s.append(float(g["betas"][0]))
        if k + 1 < total:
            dummy.step()
            sch.step()
    return lrs, b1s


class ImprovedDiffusionTrainer:
    """Drop-in for the reference class of the same name (stage-2 diffusion trainer).

    Data parallel: when a default process group exists (`torchrun ... train_3stage.py` with
    `torch.distributed.init_process_group("nccl")` called before the trainer is built - see INTEGRATION.md) every
    rank builds the same trainer; gradients are averaged by the stepper, each rank trains on its own slice of every
    batch (`ddp.ShardedLoader`), and only rank 0 writes checkpoints, TensorBoard events, sample images and the log file."""

    def __init__(self, config: Dict[str, Any], vae_checkpoint_path: str, experiment_name: str = "pokemon_diffusion",
                 components: Optional[Dict[str, Any]] = None, compute_dtype: Optional[torch.dtype] = None):
        self.config, self.vae_checkpoint_path, self.experiment_name = config, vae_checkpoint_path, experiment_name
        self._components = components or {}
        mi = config.get("mi355x", {}) or {}
        if compute_dtype is None:
            compute_dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32,
                             "float32": torch.float32}[str(mi.get("dtype", "bf16"))]
        self.compute_dtype = compute_dtype
        # the frozen VAE either side of the path runs in fp32 like the reference's (the latents the U-Net trains on and the
        # monitoring images then carry fp32-grade error, 1e-3, not bf16's 3e-2); `mi355x.vae_dtype: bf16` trades that for speed
        self.vae_dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32,
                          "float32": torch.float32}[str(mi.get("vae_dtype", "fp32"))]
        self._grad_bucket_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}[str(mi.get("grad_bucket_dtype", "fp32"))]
        # data parallel: measure the gradient-exchange mode on the first training batch (state is restored: no training happens)
        self._ddp_autotune = bool(mi.get("ddp_autotune", True))
        if not torch.cuda.is_available():
            raise _lib.PsgError("ImprovedDiffusionTrainer (MI355X build) needs a GPU; there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.rank, self.world = dist_info()
        self.is_main = self.rank == 0
        print(f"Using device: {self.device}")
        self.setup_directories()
        self.setup_logging()
        self.setup_models()
        self.setup_data_loaders()
        self.setup_optimization()
        self.setup_scheduler()
        self.setup_monitoring()
        self.current_epoch, self.best_val_loss = 0, float("inf")
        self._gs_offset = 0

    # -- setup (plain Python glue, same keys as the reference) -----------------------------------------
    def setup_directories(self):
        self.experiment_dir = Path(self.config["experiment_dir"]) / self.experiment_name
        self.checkpoint_dir, self.log_dir, self.sample_dir = (self.experiment_dir / d for d in ("checkpoints", "logs", "samples"))
        for d in (self.experiment_dir, self.checkpoint_dir, self.log_dir, self.sample_dir):
            d.mkdir(parents=True, exist_ok=True)

    def setup_logging(self):
        fmt = "%(asctime)s - %(levelname)s - %(message)s"
        logging.basicConfig(level=logging.INFO if self.is_main else logging.WARNING, format=fmt, handlers=[logging.StreamHandler()])
        self.logger = logging.getLogger(__name__)
        if self.is_main:                                 # one log file, written by rank 0 (basicConfig is a no-op when the
            path = str((self.log_dir / "diffusion_training.log").resolve())   # host application configured logging first)
            root = logging.getLogger()
            if not any(isinstance(h, logging.FileHandler) and h.baseFilename == path for h in root.handlers):
                fh = logging.FileHandler(path)
                fh.setFormatter(logging.Formatter(fmt))
                root.addHandler(fh)
            if root.level > logging.INFO or root.level == logging.NOTSET:
                root.setLevel(logging.INFO)

    def _component(self, name):
        if name in self._components:
            return self._components[name]
        if name in ("VAEEncoder", "VAEDecoder"):          # the frozen VAE either side of the path runs on the kernels too
            from . import vae
            cls = getattr(vae, name)
            return lambda *a, **k: cls(*a, compute_dtype=self.vae_dtype, **k)
        try:                                   # the reference package, when this class is dropped into its tree
            import importlib
            mod = importlib.import_module({"create_data_loaders": "src.data"}.get(name, "src.models"))
            return getattr(mod, name)
        except Exception as e:                 # noqa: BLE001
            raise _lib.PsgError(f"component '{name}' is outside the accelerated hot path: provide it via components= "
                                f"or run inside the reference tree ({e})")

    def setup_models(self):
        mc = self.config["model"]
        latent_dim = mc.get("latent_dim", 8)
        # Each collaborator is taken from `components=` when injected (an INSTANCE: `text_encoder`, `vae_encoder`,
        # `vae_decoder`), else built the reference's way (:150-225): BERT text encoder from the reference package, the frozen
        # VAE from THIS package's kernels, both loaded from the stage-1 checkpoint.  Any subset may be injected.
        comps = self._components
        ckpt = {}
        if not all(k in comps for k in ("text_encoder", "vae_encoder", "vae_decoder")):
            if os.path.exists(str(self.vae_checkpoint_path)):
                print(f"Loading VAE from {self.vae_checkpoint_path}")
                ckpt = torch.load(self.vae_checkpoint_path, map_location=self.device)
            elif "vae_encoder" not in comps:
                raise FileNotFoundError(f"VAE checkpoint {self.vae_checkpoint_path} not found (needed for the frozen VAE encoder)")
        if "text_encoder" in comps:
            self.text_encoder = comps["text_encoder"]
        else:
            TextEncoder = self._component("TextEncoder")
            self.text_encoder = TextEncoder(model_name=mc["bert_model"], hidden_dim=mc["text_embedding_dim"]).to(self.device)
            if "text_encoder_state_dict" in ckpt:
                self.text_encoder.load_state_dict(ckpt["text_encoder_state_dict"], strict=False)
        vsd = ckpt.get("vae_state_dict") if isinstance(ckpt, dict) else None
        if "vae_encoder" in comps:
            self.vae_encoder = comps["vae_encoder"]
        else:
            self.vae_encoder = self._component("VAEEncoder")(input_channels=3, latent_dim=latent_dim).to(self.device)
            if vsd is not None:
                self.vae_encoder.load_state_dict({k[8:]: v for k, v in vsd.items() if k.startswith("encoder.")}, strict=False)
        if "vae_decoder" in comps:
            self.vae_decoder = comps["vae_decoder"]
        elif "vae_encoder" in comps and vsd is None:
            self.vae_decoder = None               # (stub encoder and no stage-1 checkpoint: monitoring images are skipped)
        else:
            self.vae_decoder = self._component("VAEDecoder")(latent_dim=latent_dim, text_dim=mc["text_embedding_dim"],
                                                             output_channels=3).to(self.device)
            if vsd is not None:
                self.vae_decoder.load_state_dict({k[8:]: v for k, v in vsd.items() if k.startswith("decoder.")}, strict=False)
        for m in (self.text_encoder, self.vae_encoder, self.vae_decoder):
            if isinstance(m, torch.nn.Module):
                for p in m.parameters():
                    p.requires_grad = False
                m.eval()
        self.unet = UNet(latent_dim=latent_dim, text_dim=mc["text_embedding_dim"], time_emb_dim=mc.get("time_emb_dim", 128),
                         num_heads=mc.get("num_heads", 4), compute_dtype=self.compute_dtype).to(self.device)
        self.noise_scheduler = NoiseScheduler(num_timesteps=mc.get("num_timesteps", 1000), beta_start=mc.get("beta_start", 0.0001),
                                              beta_end=mc.get("beta_end", 0.02))
        self.logger.info(f"U-Net initialized with {sum(p.numel() for p in self.unet.parameters())} parameters")

    def setup_data_loaders(self):
        if "data_loaders" in self._components:
            loaders = dict(self._components["data_loaders"])
        else:
            dc, uc = self.config["data"], self.config.get("unet_optimization", {}) or {}
            bs, nw = uc.get("batch_size", dc["batch_size"]), uc.get("num_workers", dc["num_workers"])
            tr, va, te = self._component("create_data_loaders")(csv_path=dc["csv_path"], image_dir=dc["image_dir"], batch_size=bs,
                                                                val_split=dc["val_split"], test_split=dc["test_split"],
                                                                image_size=dc["image_size"], num_workers=nw, pin_memory=dc["pin_memory"])
            loaders = {"train": tr, "val": va, "test": te}
            self.logger.info(f"Data loaders created: train={len(tr)}, val={len(va)}, test={len(te)}")
        if self.world > 1:
            # the reference's loaders know nothing about ranks: every rank iterates the same global batches (same
            # seeding - documented requirement) and keeps its own contiguous slice of each
            loaders["train"] = ShardedLoader(loaders["train"], self.rank, self.world)
        self.data_loaders = loaders

    def setup_optimization(self):
        uc, oc = self.config.get("unet_optimization", {}) or {}, self.config["optimization"]
        get = lambda k, default=None: uc.get(k, oc.get(k, default))      # safe fallbacks (the reference KeyErrors on beta1/beta2)
        lr, self.max_grad_norm = get("learning_rate"), get("max_grad_norm")
        self._betas = (get("beta1", 0.9), get("beta2", 0.999))
        self.stepper = DiffusionStepper(self.unet, self.noise_scheduler, lr=lr, betas=self._betas,
                                        weight_decay=get("weight_decay"), eps=1e-6, max_grad_norm=self.max_grad_norm,
                                        optimizer_type=get("optimizer", "adamw"), grad_bucket_dtype=self._grad_bucket_dtype)
        self.optimizer = self.stepper.optimizer
        self.scheduler_config = {"type": uc.get("scheduler", oc.get("scheduler", "cosine")), "lr": lr}
        self.logger.info(f"Using U-Net optimization: {get('optimizer', 'adamw')}, lr={lr}, wd={get('weight_decay')}, clip={self.max_grad_norm}")

    def _make_scheduler(self, optimizer):
        if self.scheduler_config["type"] == "cosine":
            return torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=self.scheduler_config["lr"], total_steps=self._total_steps,
                                                       pct_start=0.1, anneal_strategy="cos")
        return torch.optim.lr_scheduler.ConstantLR(optimizer, factor=1.0)

    def setup_scheduler(self):
        self._total_steps = max(1, self.config["training"]["diffusion_epochs"] * len(self.data_loaders["train"]))
        if self.scheduler_config["type"] == "cosine":
            self.logger.info(f"OneCycleLR total_steps: {self._total_steps}")
        # the host-side scheduler object keeps the reference's checkpoint format ('scheduler_state_dict'); the schedule
        # the kernels follow is the same class's output, tabulated per completed-step count
        self.scheduler = self._make_scheduler(self.optimizer)
        self._fused = isinstance(self.optimizer, FusedAdamW)
        if self._fused:
            n = self._total_steps if self.scheduler_config["type"] == "cosine" else 1
            lrs, b1s = schedule_tables(self._make_scheduler, self.scheduler_config["lr"], self._betas, n)
            self._lr_host, self._b1_host = lrs, b1s
            self.optimizer.set_lr_table(lrs, b1s)

    def setup_monitoring(self):
        self.writer = _NullWriter()
        if not self.is_main:
            return
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer = SummaryWriter(log_dir=self.log_dir)
        except Exception:                      # tensorboard is optional here
            pass

    # -- step counters: the truth is on the device ---------------------------------------------------------
    @property
    def global_step(self) -> int:
        """Optimizer steps that happened (:418).  Reads the device counter: one host sync."""
        return self.stepper.steps_done() + self._gs_offset

    @global_step.setter
    def global_step(self, value: int):
        self._gs_offset = int(value) - self.stepper.steps_done()

    def _sync_host_schedule(self):
        """Bring the host-side scheduler / param_groups (logging, checkpoints) to the device step count."""
        if not self._fused:
            return
        k = self.optimizer.steps_done()
        i = min(k, len(self._lr_host) - 1)
        g = self.optimizer.param_groups[0]
        g["lr"] = self._lr_host[i]
        g["betas"] = (self._b1_host[i], g["betas"][1])
        self.scheduler.last_epoch = k
        self.scheduler._step_count = k + 1
        self.scheduler._last_lr = [g["lr"]]

    # -- facade ---------------------------------------------------------------------------------------
    def train_step(self, latents, text_emb, t, noise=None, pre_flag=None):
        """The loop body :363-413 on clean latents; see DiffusionStepper.train_step.  No host sync with the fused
        optimizer: the step count and the lr / beta1 schedule advance on the device, and only when the step happened."""
        out = self.stepper.train_step(latents, text_emb, t, noise, pre_flag=pre_flag)
        if not self._fused and (int(out["nan_flag"].item()) & FLAG_SKIP_MASK) == 0:
            self.scheduler.step()                                                                    # :413
        return out

    def sample(self, text_emb, num_samples, fast_sampling=True, noise_fn=None):
        return self.stepper.sample(text_emb, num_samples, fast_sampling, noise_fn, latent_dim=self.config["model"].get("latent_dim", 8))

    def ddpm_sample(self, text_emb: torch.Tensor, num_samples: int, fast_sampling: bool = True) -> torch.Tensor:
        return self.sample(text_emb, num_samples, fast_sampling)

    # -- epochs ----------------------------------------------------------------------------------------
    def _encode(self, batch):
        """text embedding + frozen-VAE latent (:347-360) and the device-side form of their two NaN checks."""
        images = batch["image"].to(self.device)
        with torch.no_grad():
            text_emb = self.text_encoder(batch["full_description"])
            gen = self.stepper.generator
            from .vae import VAEEncoder as _OwnEncoder
            if gen is not None and isinstance(self.vae_encoder, _OwnEncoder):      # this rank's own reparameterisation noise
                latent = self.vae_encoder(images, generator=gen)                   # (drawn with mu's shape, whatever the image size)
            elif gen is not None and images.is_cuda:
                # an injected / reference encoder draws with randn_like from the DEFAULT generator, seeded alike on every
                # rank: fork it for the call and seed the fork from this rank's stream (one host read per batch, foreign
                # encoders under data parallel only)
                with torch.random.fork_rng(devices=[images.device]):
                    torch.cuda.manual_seed(int(torch.randint(0, 2 ** 62, (1,), device=self.device, generator=gen).item()))
                    latent = self.vae_encoder(images)
            else:
                latent = self.vae_encoder(images)
            latent = latent[0] if isinstance(latent, (tuple, list)) else latent
            latent, text_emb = latent.float(), text_emb.float()
            bad = ~(torch.isfinite(text_emb).all() & torch.isfinite(latent).all())
        return latent, text_emb, bad.to(torch.int32).reshape(1) * FLAG_INPUT_BAD

    def train_epoch(self, epoch: int) -> Dict[str, float]:
        dev = self.device
        total = torch.zeros(1, device=dev)                              # sum of the losses of the batches that trained
        counts = torch.zeros(2, dtype=torch.int64, device=dev)          # [batches that trained, NaN batches (:384,392)]
        log_every = self.config["training"]["log_every"]
        for batch_idx, batch in enumerate(self.data_loaders["train"]):
            try:
                latent, text_emb, pre = self._encode(batch)
                t = self.stepper.randint_t(latent.shape[0])
                if self._ddp_autotune and self.world > 1:
                    self._ddp_autotune = False
                    rec = self.stepper.autotune_exchange(latent, text_emb, t)
                    if rec is not None and self.is_main:
                        self.logger.info(f"gradient exchange mode: {rec['chosen']} ({rec['ms_per_step']})")
                out = self.train_step(latent, text_emb, t, pre_flag=pre)
                flag = out["nan_flag"]
                good = (flag & FLAG_SKIP_MASK) == 0
                total += torch.where(good, out["loss"], torch.zeros_like(out["loss"]))
                counts[0] += good.reshape(()).to(torch.int64)
                # the reference counts a NaN batch only at :384 / :392 (prediction, loss); a bad INPUT or noisy latent
                # `continue`s earlier (:353-376) and is not counted - a NaN input also poisons the prediction, so mask it
                early = (flag & (FLAG_INPUT_BAD | FLAG_NOISY_BAD | FLAG_T_RANGE)) != 0
                counts[1] += (((flag & (FLAG_PRED_BAD | FLAG_LOSS_BAD)) != 0) & ~early).reshape(()).to(torch.int64)
                if batch_idx % log_every == 0:                      # the only host syncs of the loop
                    if int(flag.item()) & FLAG_SKIP_MASK:
                        self.logger.warning(f"NaN/Inf detected (flag {int(flag.item())}), batch skipped")
                        continue
                    self._sync_host_schedule()
                    step = self.global_step
                    self.writer.add_scalar("Diffusion Train/Loss", out["loss"].item(), step)
                    self.writer.add_scalar("Diffusion Train/Learning_Rate", self.optimizer.param_groups[0]["lr"], step)
                    self.writer.add_scalar("Diffusion Train/Gradient_Norm", out["grad_norm"].item(), step)
            except Exception as e:                                 # noqa: BLE001  (reference :433-435 logs and continues)
                self.logger.error(f"Error in training batch {batch_idx}: {e}")
                if self.world > 1:
                    # one rank skipping a batch leaves the others inside the bucket all-reduce and its own bucket counters
                    # half consumed: drop the exchange state and fail the job instead of hanging / desynchronising it
                    if self.stepper.reducer is not None:
                        self.stepper.reducer.reset()
                    raise
                continue
        nb, nan_count = (int(v) for v in counts.tolist())
        if nb == 0:
            self.logger.error("No valid batches processed!")
            return {"train_loss": float("inf")}
        avg = float(total.item()) / nb
        self._sync_host_schedule()
        self.logger.info(f"Epoch {epoch}: Average loss = {avg:.6f}, NaN batches = {nan_count}, LR = {self.optimizer.param_groups[0]['lr']:.2e}")
        return {"train_loss": avg}

    def validate_epoch(self, epoch: int) -> Dict[str, float]:
        dev = self.device
        total, nb = torch.zeros(1, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
        for batch in self.data_loaders["val"]:
            try:
                latent, text_emb, _ = self._encode(batch)          # (the reference does not check the inputs here)
                t = self.stepper.randint_t(latent.shape[0])
                loss, flag = self.stepper.eval_loss(latent, text_emb, t)
                good = (flag & (FLAG_PRED_BAD | FLAG_LOSS_BAD)) == 0                                  # :482-489
                total += torch.where(good, loss, torch.zeros_like(loss))
                nb += good.to(torch.int64)
            except Exception as e:                                 # noqa: BLE001
                self.logger.error(f"Error in validation batch: {e}")
        n = int(nb.item())
        if n == 0:
            return {"val_loss": float("inf")}
        avg = float(total.item()) / n
        self.writer.add_scalar("Diffusion Val/Loss", avg, epoch)
        return {"val_loss": avg}

    def generate_samples(self, epoch: int, num_samples: int = 8):
        """Monitoring only (:571-615): needs the (out-of-scope) VAE decoder; skipped when it is absent; rank 0 only."""
        if self.vae_decoder is None or not self.is_main:
            return
        batch = next(iter(self.data_loaders["val"]))
        desc = batch["full_description"][:num_samples]
        with torch.no_grad():
            text_emb = self.text_encoder(desc).float()
            for i in range(0, len(desc), 4):
                te = text_emb[i:i + 4]
                imgs = self.vae_decoder(self.ddpm_sample(te, te.shape[0]), te)
                imgs = torch.clamp((imgs + 1.0) / 2.0, 0, 1)
                for j, img in enumerate(imgs):
                    self.writer.add_image(f"Diffusion Generated/Sample_{i + j}", img.cpu(), epoch)
        self.logger.info(f"Generated {len(desc)} samples for epoch {epoch}")

    # -- checkpoints (wire format of :617-655) ------------------------------------------------------------
    def save_checkpoint(self, epoch: int, is_best: bool = False):
        if not self.is_main:                       # replicas are identical: one writer
            return
        self._sync_host_schedule()
        ckpt = {"epoch": epoch, "global_step": self.global_step, "unet_state_dict": self.unet.state_dict(),
                "optimizer_state_dict": self.optimizer.state_dict(), "scheduler_state_dict": self.scheduler.state_dict(),
                "best_val_loss": self.best_val_loss, "config": self.config}
        if is_best:
            torch.save(ckpt, self.checkpoint_dir / "diffusion_best_model.pth")
            self.logger.info(f"New best model saved at epoch {epoch}")

    def load_checkpoint(self, checkpoint_path: str):
        ckpt = torch.load(checkpoint_path, map_location=self.device)
        self.current_epoch, self.best_val_loss = ckpt["epoch"], ckpt["best_val_loss"]
        self.unet.load_state_dict(ckpt["unet_state_dict"])
        self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        if self.scheduler and ckpt["scheduler_state_dict"]:
            self.scheduler.load_state_dict(ckpt["scheduler_state_dict"])
        if not self._fused:
            self.stepper.host_steps = int(ckpt["scheduler_state_dict"].get("last_epoch", 0)) if ckpt["scheduler_state_dict"] else 0
        self.global_step = ckpt["global_step"]
        from .ops import WeightCache
        WeightCache.invalidate()
        self.logger.info(f"Checkpoint loaded from {checkpoint_path}")

    def train(self):
        self.logger.info("Starting diffusion training...")
        tc = self.config["training"]
        for epoch in range(self.current_epoch, tc["diffusion_epochs"]):
            self.current_epoch = epoch
            tm = self.train_epoch(epoch)
            if tm["train_loss"] == float("inf"):
                self.logger.error(f"Training failed at epoch {epoch}, stopping")
                break
            vm = self.validate_epoch(epoch)
            if epoch % tc["sample_every"] == 0:
                self.generate_samples(epoch)
            is_best = vm["val_loss"] < self.best_val_loss
            if is_best:
                self.best_val_loss = vm["val_loss"]
            if epoch % tc["save_every"] == 0 or is_best:
                self.save_checkpoint(epoch, is_best)
            self.logger.info(f"Epoch {epoch}: train_loss={tm['train_loss']:.4f}, val_loss={vm['val_loss']:.4f}")
        self.logger.info("diffusion training completed!")
        self.writer.close()


DiffusionTrainer = ImprovedDiffusionTrainer     # src/training/__init__.py:7 and train_3stage.py:19 alias it this way

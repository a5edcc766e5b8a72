"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The per-image train step shards along the batch (every op is per-sample; only
the loss mean and the weight gradients reduce across samples), so the only
collective is ONE averaged all-reduce of the flat gradient arena per step, issued
in large buckets (default 64 MiB of fp32) on a side stream as soon as the
autograd engine has produced every gradient of a bucket, overlapping the rest
of backward.  xGMI is point-to-point, so few large messages beat many small ones.
Backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests of the bucket logic.

`bucket_dtype=torch.bfloat16` halves the bytes on the links (1.28 GB instead of 2.56 GB per step): each bucket is
rounded to bf16 into a staging buffer, summed by RCCL in bf16 and widened back into the fp32 arena.  That is for the
strong-scaling corner (per-GPU batch <= 32) where a single ring's 29 ms would not hide behind backward
(SURVEY.md §8e); at per-GPU batch 256 the fp32 exchange hides completely and stays the default.
"""
import os
import time

import torch
import torch.distributed as dist

# CUs the tile choosers leave to an overlapped all-reduce (tools/contention.py): a launch planned as ONE round of
# workgroups over 256 CUs takes two rounds as soon as a collective's channels sit on a few of them (batch-256 step on a
# CU-masked stream: 240 CUs +26 %, 224 CUs +28 % with plans made for 256; +11 % at 224 with plans made for 224).  The
# dispatcher deals workgroups round-robin over 8 XCDs x 4 shader engines, so the reserve is a whole CU per engine (32), and
# RCCL is capped at that many channels (`configure_rccl`, before init_process_group).
CU_RESERVE = int(os.environ.get("PSG_DDP_CU_RESERVE", "32"))
# ... and which launches plan around it: those of at most this many rounds of workgroups on the whole chip (0: all).  A one-round
# grid takes twice as long when CUs are missing, a ten-round grid 14 % longer whatever the plan - and planning it for 224 CUs
# costs those 14 % also while the CUs are free (at world size 1, nothing taken: +5.6 % of the step with every launch planned
# for 224 CUs).
RESERVE_ROUNDS = int(os.environ.get("PSG_DDP_RESERVE_ROUNDS", "3"))


def configure_rccl(max_channels=None):
    """Cap the channels (= resident workgroups, one CU each) RCCL may use at the CU reserve the tile choosers plan for.
    Call BEFORE torch.distributed.init_process_group; an explicit NCCL_MAX_NCHANNELS in the environment wins."""
    n = CU_RESERVE if max_channels is None else int(max_channels)
    if n > 0:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(n))
    return int(os.environ.get("NCCL_MAX_NCHANNELS", "0") or 0)


def dist_info():
    """(rank, world size) of the default process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def rank_generator(device, rank=None, world=None):
    """Generator for the per-sample random draws of a data-parallel rank (timesteps, noise, the VAE's reparameterisation
    eps), or None (= torch's default generator) in a single process.  The reference's loaders seed every process alike
    (dataset_improved.py:254), which the loader shuffle NEEDS (ddp.ShardedLoader) - but the same seed on every device
    generator would hand all N shards of a global batch identical t / noise / eps: B/N distinct draws instead of B.
    Seed: torch.initial_seed() mixed with the rank by a golden-ratio multiply (distinct, well-separated streams; the CPU
    generator stays shared, so shuffles still agree)."""
    if rank is None or world is None:
        rank, world = dist_info()
    if world <= 1:
        return None
    g = torch.Generator(device=device)
    g.manual_seed((torch.initial_seed() + 0x9E3779B97F4A7C15 * (int(rank) + 1)) & 0x7FFFFFFFFFFFFFFF)
    return g


def _digest(batch):
    """Order-sensitive 64-bit digest of a batch's tensors and strings (ShardedLoader's agreement check)."""
    import zlib
    acc = 0
    vals = batch.values() if isinstance(batch, dict) else (batch if isinstance(batch, (list, tuple)) else [batch])
    for v in vals:
        if torch.is_tensor(v):
            b = v.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes() if v.numel() else b""
        elif isinstance(v, (list, tuple)):
            b = "\x1f".join(str(e) for e in v).encode()
        else:
            b = str(v).encode()
        acc = (acc * 1000003 + zlib.crc32(b)) & 0x7FFFFFFFFFFFFFFF
    return acc


class ShardedLoader:
    """Rank r's view of a loader that yields GLOBAL batches: the contiguous slice [r*n/N, (r+1)*n/N) of every batch
    (SURVEY.md §8e), for loaders that were built without a DistributedSampler (the reference's `create_data_loaders`).
    All ranks must iterate the same batches in the same order (same seed / shuffle state on every rank); a ragged last
    batch is trimmed to a multiple of the world size so every rank steps the same number of times with equal weight.
    Works on dict batches (tensors, lists, tuples sliced along dim 0), tensors and tuples/lists of tensors.

    `verify` (default: whenever a process group exists): the FIRST global batch of every epoch is digested on each rank
    and the digests are compared through one tiny all-reduce (MIN and MAX of the digest); ranks that shuffled differently
    would otherwise train silently on overlapping / incoherent shards - here they raise on every rank at once."""

    def __init__(self, loader, rank, world, verify=None, group=None):
        self.loader, self.rank, self.world = loader, int(rank), int(world)
        self.verify = (dist.is_available() and dist.is_initialized() and self.world > 1) if verify is None else bool(verify)
        self.group = group

    def _check_agreement(self, batch):
        d = _digest(batch)
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        lo = torch.tensor([d], dtype=torch.int64, device=dev)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if int(lo.item()) != int(hi.item()):
            raise RuntimeError("ShardedLoader: ranks iterate DIFFERENT global batches (first-batch digests disagree): the loaders "
                               "must be seeded alike on every rank (same shuffle state) - see INTEGRATION.md")

    def __len__(self):
        return len(self.loader)

    def _slice(self, v, lo, hi, n):
        if torch.is_tensor(v):
            return v[lo:hi] if v.dim() > 0 and v.shape[0] == n else v
        if isinstance(v, (list, tuple)) and len(v) == n:
            return type(v)(v[lo:hi]) if isinstance(v, tuple) else v[lo:hi]
        return v

    @staticmethod
    def _batch_size(batch):
        vals = batch.values() if isinstance(batch, dict) else (batch if isinstance(batch, (list, tuple)) else [batch])
        for v in vals:
            if torch.is_tensor(v) and v.dim() > 0:
                return v.shape[0]
        for v in vals:
            if isinstance(v, (list, tuple)):
                return len(v)
        raise TypeError("ShardedLoader: cannot find the batch dimension")

    def __iter__(self):
        first = True
        for batch in self.loader:
            if first and self.verify:
                self._check_agreement(batch)
            first = False
            n = self._batch_size(batch)
            per = n // self.world
            if per == 0:                     # fewer samples than ranks: every rank skips it together
                continue
            lo, hi = self.rank * per, (self.rank + 1) * per
            if isinstance(batch, dict):
                yield {k: self._slice(v, lo, hi, n) for k, v in batch.items()}
            elif isinstance(batch, (list, tuple)):
                yield type(batch)(self._slice(v, lo, hi, n) for v in batch)
            else:
                yield self._slice(batch, lo, hi, n)


class BucketedAllReduce:
    """`force_single`: run the collectives even in a group of ONE rank (the exchange is the identity there).  That is how
    the RCCL code path - communicator, ncclAvg, bf16 buckets, async work handles on the side stream - executes on a 1-GPU
    box: `DiffusionStepper(distributed="force")`, `bench.py --gpus 1 --force-ddp`, tests/test_ddp_gpu.py."""

    def __init__(self, arena_flat, params, offsets, bucket_bytes=64 << 20, process_group=None, overlap=True,
                 bucket_dtype=torch.float32, force_single=False, cu_reserve=None):
        self.flat = arena_flat
        self.cu_reserve = CU_RESERVE if cu_reserve is None else int(cu_reserve)
        self.reserve_rounds = RESERVE_ROUNDS
        self._cus_set = False
        self.tuned = None                  # autotune()'s record
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (bool(force_single) and dist.is_initialized())
        self.overlap = overlap and self.flat.is_cuda
        if bucket_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("BucketedAllReduce: bucket_dtype must be float32 or bfloat16")
        self.bucket_dtype = bucket_dtype
        # RCCL averages inside the collective (ncclAvg): no extra pass over the 2.56 GB arena; gloo sums, then scales
        self.avg_in_collective = self._avg_capability(process_group)
        self.op = dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM
        self.buckets = []          # (start, end, [param indices])
        cap = max(1, bucket_bytes // 4)
        start, members = 0, []
        order = sorted(range(len(offsets)), key=lambda i: offsets[i])            # memory order (an arena may use its own layout)
        for pos, i in enumerate(order):
            end = offsets[order[pos + 1]] if pos + 1 < len(order) else self.flat.numel()      # (includes the alignment gap)
            members.append(i)
            if end - start >= cap:
                self.buckets.append((start, end, members))
                start, members = end, []
        if members:
            self.buckets.append((start, self.flat.numel(), members))
        self._bucket_of = {}
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self._bucket_of[i] = b
        self._pending = [0] * len(self.buckets)
        self._works = []
        self._stream = torch.cuda.Stream() if self.overlap else None
        self._staging = None
        if bucket_dtype != torch.float32:
            self._staging = torch.empty(self.flat.numel(), dtype=bucket_dtype, device=self.flat.device)
        self.bytes_per_step = self.flat.numel() * (2 if bucket_dtype == torch.bfloat16 else 4)
        self.launched_early = 0            # buckets of the LAST finished step that left before finish() (overlap evidence)
        self._early = 0
        self.reset()

    def _avg_capability(self, group):
        """ncclAvg or sum-then-scale - decided WITHOUT a trial collective (an exception on one rank only would leave the
        others inside it): the local answer comes from the library version (ncclAvg exists since NCCL 2.10; RCCL reports
        its NCCL API level), and the ranks then agree on the MINIMUM of their answers through a plain SUM-free MIN
        all-reduce that every backend supports, so a mixed installation falls back together."""
        if not (dist.is_available() and dist.is_initialized()) or not self.active or dist.get_backend(group) != "nccl":
            return False
        try:
            ver = torch.cuda.nccl.version()
            ver = tuple(ver) if isinstance(ver, (tuple, list)) else (ver // 1000, (ver // 100) % 10, ver % 100)
            local = 1 if tuple(ver[:2]) >= (2, 10) else 0
        except Exception:                          # noqa: BLE001 - no version query: assume the old API
            local = 0
        agree = torch.tensor([local], dtype=torch.int32, device=self.flat.device)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN, group=group)
        return bool(int(agree.item()))

    def autotune(self, run_step, trials=3):
        """Choose HOW the exchange runs on this machine by measuring it (the answer depends on how many CUs RCCL's channels
        take and on the links - unknowable on a 1-GPU box): `run_step()` executes one full train step; four settings are
        timed, `trials` steps each after one untimed step: buckets overlapped with backward and EVERY launch planned around the CU
        reserve ("overlap+reserve"), only the launches of few rounds planned around it ("overlap+reserve-few-rounds"),
        overlapped with plans for the whole chip ("overlap"), and the whole exchange after backward ("deferred": no contention,
        nothing hidden).  Ranks agree through a MAX all-reduce of the times (the job runs at the
        pace of its slowest rank) and all take the same fastest setting.  Returns the record (also kept in `.tuned`)."""
        if not self.active:
            return None
        rr = self.reserve_rounds if self.reserve_rounds > 0 else 3
        settings = [("overlap+reserve", True, self.cu_reserve, 0), ("overlap+reserve-few-rounds", True, self.cu_reserve, rr),
                    ("overlap", True, 0, 0), ("deferred", False, 0, 0)]
        if self.cu_reserve <= 0 or self._stream is None:
            settings = [x for x in settings if not x[0].startswith("overlap+reserve")]
        if self._stream is None:
            settings = [x for x in settings if x[1] is False]
        times = []
        start = (self.overlap, self.cu_reserve, self.reserve_rounds)
        try:
            for name, ov, res, rounds in settings:
                self.overlap, self.cu_reserve, self.reserve_rounds = ov and self._stream is not None, res, rounds
                run_step()
                if self.flat.is_cuda:
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(trials):
                    run_step()
                if self.flat.is_cuda:
                    torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) / trials)
        except BaseException:
            # a step failed mid-measurement: leave the reducer as it was configured (and the planners on the whole chip)
            self.overlap, self.cu_reserve, self.reserve_rounds = start
            self.reset()
            if self._cus_set:
                self._plan_cus(0)
            raise
        tt = torch.tensor(times, dtype=torch.float64, device=self.flat.device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=self.group)
        tl = [float(x) for x in tt.tolist()]
        best = min(range(len(tl)), key=lambda i: tl[i])
        name, ov, res, rounds = settings[best]
        self.overlap, self.cu_reserve, self.reserve_rounds = ov and self._stream is not None, res, rounds
        self.tuned = {"chosen": name, "ms_per_step": {settings[i][0]: 1e3 * tl[i] for i in range(len(tl))}, "trials": trials}
        return self.tuned

    def reset(self):
        """Back to 'no gradient seen' (start of a step; also the recovery point after a failed step: pending counters and
        outstanding work handles of a half-finished exchange are dropped)."""
        for b, (_, _, mem) in enumerate(self.buckets):
            self._pending[b] = len(mem)
        self._works = []
        self._early = 0

    def on_ready(self, i):
        """Called (by the gradient sink) when parameter i's gradient is final for this step."""
        if not self.active or not self.overlap:
            return
        b = self._bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._early += 1
            self._launch(b)

    def _reduce(self, s, e):
        view = self.flat[s:e]
        if self._staging is None:
            return dist.all_reduce(view, op=self.op, group=self.group, async_op=True), None
        st = self._staging[s:e]
        st.copy_(view)                                             # fp32 -> bf16 (round to nearest even)
        return dist.all_reduce(st, op=self.op, group=self.group, async_op=True), (view, st)

    def _plan_cus(self, reserve):
        """Tell the tile choosers how many CUs they have while buckets are in flight (host-side planning input; the host runs
        ahead of the GPU by a few launches, so the window is approximate - a plan for 224 CUs on 256 free ones costs ~1 %)."""
        if not self.flat.is_cuda:
            return
        from . import _lib
        _lib.set_available_cus(self.flat.device.index if self.flat.device.index is not None else torch.cuda.current_device(),
                               0 if reserve <= 0 else 256 - int(reserve), self.reserve_rounds if reserve > 0 else 0)
        self._cus_set = reserve > 0

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        if self.overlap and self.cu_reserve > 0 and not self._cus_set and self._stream is not None:
            self._plan_cus(self.cu_reserve)          # first bucket of the step leaves during backward: plan around it from here on
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            from .ops import SideStream            # weight gradients are produced on the side stream
            if SideStream.enabled and self.flat.is_cuda:
                self._stream.wait_stream(SideStream.get(self.flat.device))
            with torch.cuda.stream(self._stream):
                w = self._reduce(s, e)
        else:
            w = self._reduce(s, e)
        self._works.append(w)

    def finish(self):
        """Complete the exchange: launch buckets not yet sent, wait, and average."""
        if not self.active:
            return
        for b in range(len(self.buckets)):
            if self._pending[b] != 0 or not self.overlap:
                if self._pending[b] >= 0:
                    self._launch(b)
            self._pending[b] = -1
        for w, back in self._works:
            if back is None:
                w.wait()                                           # (the CURRENT stream waits for the collective)
            elif self._stream is not None:                         # widen the summed bf16 bucket back into the arena
                with torch.cuda.stream(self._stream):
                    w.wait()
                    back[0].copy_(back[1])
            else:
                w.wait()
                back[0].copy_(back[1])
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        if self._cus_set:
            self._plan_cus(0)                        # exchange over: the whole chip again (clip, AdamW, next forward)
        if not self.avg_in_collective:
            self.flat.mul_(1.0 / self.world)
        early = self._early
        self.reset()
        self.launched_early = early

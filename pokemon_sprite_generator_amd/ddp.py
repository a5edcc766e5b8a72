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
import torch
import torch.distributed as dist


def dist_info():
    """(rank, world size) of the default process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class ShardedLoader:
    """Rank r's view of a loader that yields GLOBAL batches: the contiguous slice [r*n/N, (r+1)*n/N) of every batch
    (SURVEY.md §8e), for loaders that were built without a DistributedSampler (the reference's `create_data_loaders`).
    All ranks must iterate the same batches in the same order (same seed / shuffle state on every rank); a ragged last
    batch is trimmed to a multiple of the world size so every rank steps the same number of times with equal weight.
    Works on dict batches (tensors, lists, tuples sliced along dim 0), tensors and tuples/lists of tensors."""

    def __init__(self, loader, rank, world):
        self.loader, self.rank, self.world = loader, int(rank), int(world)

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
        for batch in self.loader:
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
    def __init__(self, arena_flat, params, offsets, bucket_bytes=64 << 20, process_group=None, overlap=True,
                 bucket_dtype=torch.float32):
        self.flat = arena_flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap and self.flat.is_cuda
        if bucket_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("BucketedAllReduce: bucket_dtype must be float32 or bfloat16")
        self.bucket_dtype = bucket_dtype
        # RCCL averages inside the collective (ncclAvg): no extra pass over the 2.56 GB arena; gloo sums, then scales
        self.avg_in_collective = False
        if dist.is_initialized() and self.world > 1 and dist.get_backend(process_group) == "nccl":
            try:                                   # every rank runs the same probe, so a refusal is collective too
                probe = torch.ones(1, dtype=torch.float32, device=self.flat.device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=process_group)
                self.avg_in_collective = abs(float(probe.item()) - 1.0) < 1e-6
            except Exception:                      # noqa: BLE001 - older RCCL without ncclAvg: sum, then scale
                self.avg_in_collective = False
        self.op = dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM
        self.buckets = []          # (start, end, [param indices])
        cap = max(1, bucket_bytes // 4)
        start, members = 0, []
        for i, (p, o) in enumerate(zip(params, offsets)):
            end = offsets[i + 1] if i + 1 < len(offsets) else self.flat.numel()      # (includes the alignment gap)
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
        self.reset()

    def reset(self):
        for b, (_, _, mem) in enumerate(self.buckets):
            self._pending[b] = len(mem)
        self._works = []

    def on_ready(self, i):
        """Called (by the gradient sink) when parameter i's gradient is final for this step."""
        if self.world == 1 or not self.overlap:
            return
        b = self._bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _reduce(self, s, e):
        view = self.flat[s:e]
        if self._staging is None:
            return dist.all_reduce(view, op=self.op, group=self.group, async_op=True), None
        st = self._staging[s:e]
        st.copy_(view)                                             # fp32 -> bf16 (round to nearest even)
        return dist.all_reduce(st, op=self.op, group=self.group, async_op=True), (view, st)

    def _launch(self, b):
        s, e, _ = self.buckets[b]
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
        if self.world == 1:
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
        if not self.avg_in_collective:
            self.flat.mul_(1.0 / self.world)
        self.reset()

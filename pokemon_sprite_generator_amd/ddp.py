"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The per-image train step shards along the batch (every op is per-sample; only
the loss mean and the weight gradients reduce across samples), so the only
collective is ONE averaged all-reduce of the flat gradient arena per step, issued
in large buckets (default 64 MiB of fp32) on a side stream as soon as the
autograd engine has produced every gradient of a bucket, overlapping the rest
of backward.  xGMI is point-to-point, so few large messages beat many small ones.
Backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests of the bucket logic.
"""
import torch
import torch.distributed as dist


class BucketedAllReduce:
    def __init__(self, arena_flat, params, offsets, bucket_bytes=64 << 20, process_group=None, overlap=True):
        self.flat = arena_flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap and self.flat.is_cuda
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

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        view = self.flat[s:e]
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            from .ops import SideStream            # weight gradients are produced on the side stream
            if SideStream.enabled and self.flat.is_cuda:
                self._stream.wait_stream(SideStream.get(self.flat.device))
            with torch.cuda.stream(self._stream):
                w = dist.all_reduce(view, op=self.op, group=self.group, async_op=True)
        else:
            w = dist.all_reduce(view, op=self.op, group=self.group, async_op=True)
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
        for w in self._works:
            w.wait()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        if not self.avg_in_collective:
            self.flat.mul_(1.0 / self.world)
        self.reset()

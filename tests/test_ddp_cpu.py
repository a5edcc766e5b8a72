"""CPU suite: the N>1 gradient exchange (bucketed all-reduce of the flat gradient arena) with
world_size=2 over gloo.  The data path has exactly one collective; sharding is along the batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bucket_bytes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pokemon_sprite_generator_amd.ddp import BucketedAllReduce
        torch.manual_seed(0)
        shapes = [(64, 3, 3, 3), (64,), (130, 7), (5,), (1000, 33), (9,)]
        params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        offsets, off = [], 0
        for p in params:
            offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(off)
        for p, o in zip(params, offsets):
            p.grad = flat[o:o + p.numel()].view_as(p)
        red = BucketedAllReduce(flat, params, offsets, bucket_bytes=bucket_bytes, overlap=False)
        assert red.world == world
        for step in range(2):
            for i, p in enumerate(params):
                p.grad.copy_(torch.full(p.shape, float((rank + 1) * (i + 1) + step)))
            red.finish()
            for i, p in enumerate(params):
                want = sum((r + 1) * (i + 1) + step for r in range(world)) / world
                assert torch.allclose(p.grad, torch.full(p.shape, want)), (rank, i, step)
        q.put((rank, len(red.buckets), "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, -1, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [1 << 10, 64 << 20])
def test_bucketed_allreduce_gloo_world2(bucket_bytes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, nb, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
    if bucket_bytes == 1 << 10:
        assert res[0][1] > 1          # several buckets
    else:
        assert res[0][1] == 1         # everything in one bucket


def test_batch_sharding_is_exact_for_mean_losses():
    """Averaging equal-shard gradients reproduces the global-batch gradient (why DP needs no other collective)."""
    torch.manual_seed(0)
    w = torch.randn(5, 3, requires_grad=True)
    x, y = torch.randn(8, 3), torch.randn(8, 5)
    full = torch.autograd.grad(torch.nn.functional.smooth_l1_loss(x @ w.t(), y, beta=0.1), w)[0]
    parts = [torch.autograd.grad(torch.nn.functional.smooth_l1_loss(x[i:i + 4] @ w.t(), y[i:i + 4], beta=0.1), w)[0] for i in (0, 4)]
    assert torch.allclose(full, (parts[0] + parts[1]) / 2, atol=1e-6)

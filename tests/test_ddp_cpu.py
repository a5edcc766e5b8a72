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


def _worker(rank, world, port, bucket_bytes, q, bucket_dtype="fp32"):
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
        red = BucketedAllReduce(flat, params, offsets, bucket_bytes=bucket_bytes, overlap=False,
                                bucket_dtype=torch.bfloat16 if bucket_dtype == "bf16" else torch.float32)
        assert red.world == world
        assert red.bytes_per_step == flat.numel() * (2 if bucket_dtype == "bf16" else 4)
        for step in range(2):
            for i, p in enumerate(params):
                p.grad.copy_(torch.full(p.shape, float((rank + 1) * (i + 1) + step)))
            red.finish()
            for i, p in enumerate(params):
                want = sum((r + 1) * (i + 1) + step for r in range(world)) / world
                assert torch.allclose(p.grad, torch.full(p.shape, want)), (rank, i, step)
        # the start-up measurement that picks the exchange mode: ranks must leave it with the SAME choice (MAX all-reduce of the
        # times).  On CPU tensors there is no side stream, so "deferred" is the only candidate - the agreement path still runs.
        calls = []

        def one_step():
            calls.append(1)
            for i, p in enumerate(params):
                p.grad.copy_(torch.full(p.shape, float(rank + i)))
            red.finish()
        tuned = red.autotune(one_step, trials=2)
        assert tuned["chosen"] == "deferred" and list(tuned["ms_per_step"]) == ["deferred"] and len(calls) == 3, tuned
        assert red.overlap is False and red.tuned is tuned
        q.put((rank, len(red.buckets), "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, -1, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("bucket_bytes", [1 << 10, 64 << 20])
def test_bucketed_allreduce_gloo_world2(bucket_bytes, bucket_dtype):
    """(the test values are small integers and halves: exact in bf16 too)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, q, bucket_dtype)) for r in range(2)]
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


def test_sharded_loader_slices_every_global_batch():
    """ddp.ShardedLoader: rank r keeps rows [r*n/N, (r+1)*n/N) of every batch the wrapped loader yields (dict batches
    of tensors and string lists, like the reference's `create_data_loaders`); ragged tails are trimmed equally."""
    from pokemon_sprite_generator_amd.ddp import ShardedLoader
    batches = [{"image": torch.arange(8 * 3).reshape(8, 3), "full_description": [f"d{i}" for i in range(8)], "meta": 5},
               {"image": torch.arange(5 * 3).reshape(5, 3), "full_description": [f"e{i}" for i in range(5)], "meta": 6},
               {"image": torch.arange(1 * 3).reshape(1, 3), "full_description": ["f0"], "meta": 7}]
    got = [list(ShardedLoader(batches, r, 2)) for r in range(2)]
    assert len(got[0]) == len(got[1]) == 2                      # the 1-sample batch is dropped on every rank
    assert len(ShardedLoader(batches, 0, 2)) == 3               # (len is the wrapped loader's: the schedule length)
    for r in range(2):
        assert torch.equal(got[r][0]["image"], batches[0]["image"][4 * r:4 * r + 4])
        assert got[r][0]["full_description"] == [f"d{i}" for i in range(4 * r, 4 * r + 4)]
        assert got[r][0]["meta"] == 5
        assert torch.equal(got[r][1]["image"], batches[1]["image"][2 * r:2 * r + 2])     # 5 -> 2 + 2, last one trimmed
        assert got[r][1]["full_description"] == [f"e{i}" for i in range(2 * r, 2 * r + 2)]
    tup = [(torch.arange(6), torch.arange(6) * 2)]
    a, b = list(ShardedLoader(tup, 1, 3))[0]
    assert a.tolist() == [2, 3] and b.tolist() == [4, 6]


def test_schedule_tables_reproduce_onecycle():
    """trainer.schedule_tables: entry k = (lr, beta1) the reference's scheduler holds after k scheduler.step() calls
    (OneCycleLR cycles Adam's beta1 between 0.95 and 0.85, improved_diffusion_trainer.py:313-319)."""
    from pokemon_sprite_generator_amd.trainer import schedule_tables
    mk = lambda opt: torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-4, total_steps=20, pct_start=0.1, anneal_strategy="cos")
    lrs, b1s = schedule_tables(mk, 3e-4, (0.9, 0.999), 20)
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=3e-4, betas=(0.9, 0.999))
    sch = mk(opt)
    assert len(lrs) == 20
    for k in range(20):
        assert lrs[k] == opt.param_groups[0]["lr"] and b1s[k] == opt.param_groups[0]["betas"][0], k
        if k < 19:
            opt.step(); sch.step()
    assert abs(lrs[0] - 3e-4 / 25) < 1e-12 and abs(b1s[0] - 0.95) < 1e-12 and abs(max(lrs) - 3e-4) < 1e-12 and abs(min(b1s) - 0.85) < 1e-12
    lrs, b1s = schedule_tables(lambda o: torch.optim.lr_scheduler.ConstantLR(o, factor=1.0), 1e-4, (0.9, 0.999), 1)
    assert lrs == [1e-4] and b1s == [0.9]


def test_arena_displacement_is_loud():
    """A second ParamArena / GradArena over the same parameters displaces the first: the older ones raise instead of
    running on re-bound storage / cleared gradient sinks; unrelated arenas are untouched."""
    from pokemon_sprite_generator_amd.optim import ArenaDisplaced, GradArena, ParamArena
    from pokemon_sprite_generator_amd.ops import GradSink
    ps = [torch.nn.Parameter(torch.randn(4, 3, 3, 3)), torch.nn.Parameter(torch.randn(7))]
    other = [torch.nn.Parameter(torch.randn(5))]
    pa1, ga1, gb = ParamArena(ps), GradArena(ps), GradArena(other)
    w = ps[0].detach().clone()
    pa2, ga2 = ParamArena(ps), GradArena(ps)
    assert torch.equal(ps[0].detach(), w)                        # values survive the re-binding
    assert ps[0].data_ptr() >= pa2.flat.data_ptr() and ps[0].data_ptr() < pa2.flat.data_ptr() + pa2.flat.numel() * 4
    with pytest.raises(ArenaDisplaced):
        pa1.check_alive()
    with pytest.raises(ArenaDisplaced):
        ga1.zero()
    ga2.zero(); gb.zero()                                        # the newest and the unrelated one still work
    assert GradSink.get(ps[0]).owner is ga2 and GradSink.get(other[0]).owner is gb
    ga2.release()
    assert GradSink.get(ps[0]) is None and GradSink.get(other[0]) is not None
    gb.release()


def _rng_worker(rank, world, port, q, same_shuffle):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pokemon_sprite_generator_amd.ddp import ShardedLoader, rank_generator
        torch.manual_seed(1234)                       # the reference seeds every process alike (dataset_improved.py:254)
        gen = rank_generator(torch.device("cpu"))
        t = torch.randint(0, 1000, (16,), generator=gen)
        noise = torch.randn(4, 8, generator=gen)
        shuffle = torch.randperm(12)                  # default generator: must still agree across ranks
        # every rank iterates the same global batches and keeps its own slice; a rank that shuffled differently is caught
        order = shuffle if same_shuffle or rank == 0 else torch.flip(shuffle, dims=[0])
        data = torch.arange(12 * 3).reshape(12, 3)[order]
        batches = [{"image": data[i:i + 4], "full_description": [f"s{int(v)}" for v in order[i:i + 4]]} for i in range(0, 12, 4)]
        err, mine = None, None
        try:
            mine = [b["image"].clone() for b in ShardedLoader(batches, rank, world)]
        except RuntimeError as e:
            err = str(e)
        q.put((rank, t, noise, shuffle, mine, err))
    except Exception as e:  # noqa: BLE001
        q.put((rank, None, None, None, None, "worker: " + repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("same_shuffle", [True, False])
def test_ranks_draw_their_own_t_and_noise_but_share_the_shuffle(same_shuffle):
    """with every process seeded alike, all ranks used to draw identical timesteps / noise for their
    different shards.  `ddp.rank_generator` gives each rank its own stream for those draws while the default generator (the
    loader shuffle) stays shared; `ShardedLoader` verifies that the ranks really iterate the same global batches."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rng_worker, args=(r, 2, port, q, same_shuffle)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    (r0, t0, n0, s0, m0, e0), (r1, t1, n1, s1, m1, e1) = res
    assert t0 is not None and t1 is not None, (e0, e1)
    assert not torch.equal(t0, t1) and not torch.equal(n0, n1)          # per-rank draws differ ...
    assert torch.equal(s0, s1)                                            # ... the shared CPU stream does not
    if same_shuffle:
        assert e0 is None and e1 is None
        for a, b in zip(m0, m1):                                         # disjoint halves of the same global batch
            assert a.shape == b.shape == (2, 3) and not torch.equal(a, b)
    else:
        assert e0 is not None and e1 is not None and "DIFFERENT global batches" in e0 and "DIFFERENT global batches" in e1


def test_rank_generator_single_process_is_the_default_stream():
    from pokemon_sprite_generator_amd.ddp import rank_generator
    assert rank_generator(torch.device("cpu")) is None                   # world 1: torch's default generator (reference behaviour)
    a, b = rank_generator(torch.device("cpu"), 0, 4), rank_generator(torch.device("cpu"), 3, 4)
    assert a.initial_seed() != b.initial_seed()


def test_arena_layout_and_memory_order_buckets():
    """An arena may place parameters in its own memory order (UNet.arena_layout: the 17 time_proj / text_proj weights adjacent
    so that they are one GEMM operand); offsets stay indexed by PARAMETER, and the all-reduce buckets follow the memory order:
    they tile the flat buffer and each parameter lies wholly inside its bucket."""
    from pokemon_sprite_generator_amd.ddp import BucketedAllReduce
    from pokemon_sprite_generator_amd.optim import _arena_offsets
    shapes = [(64, 9), (64,), (130, 7), (8,), (1000, 33), (16,)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    layout = [4, 1, 3, 5, 0, 2]
    offsets, numel = _arena_offsets(params, layout)
    assert offsets[4] == 0 and offsets[1] == 33000 and offsets[3] == 33064              # 8-element steps, layout order
    assert _arena_offsets(params)[0] == [0, 576, 640, 1552, 1560, 34560]                # default: parameter order
    with pytest.raises(ValueError):
        _arena_offsets(params, [0, 1, 2, 3, 4, 4])
    flat = torch.zeros(numel)
    red = BucketedAllReduce(flat, params, offsets, bucket_bytes=1 << 12, overlap=False)
    spans = sorted((s, e) for s, e, _ in red.buckets)
    assert spans[0][0] == 0 and spans[-1][1] == numel and all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
    for b, (s, e, mem) in enumerate(red.buckets):
        for i in mem:
            assert s <= offsets[i] and offsets[i] + params[i].numel() <= e and red._bucket_of[i] == b
    assert sorted(i for _, _, mem in red.buckets for i in mem) == list(range(len(params)))

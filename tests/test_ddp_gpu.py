"""-m gpu: the data-parallel train step with 2 ranks (one process each, both on cuda:0, gloo transport so
it runs on a 1-GPU box; the production backend is RCCL - the same test runs with backend "nccl", one rank per GPU,
whenever the box has >= 2 GPUs).  Exercises the real path: gradient sink ->
bucket-ready callbacks -> async all-reduce on the side stream -> finish() -> clip -> AdamW, and checks that
2 ranks x batch 2 reproduce a single process at batch 4 (loss mean and averaged gradients)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=0):
    import pokemon_sprite_generator_amd as psg
    torch.manual_seed(seed)
    # a real (small) slice of the network: one attention UNetBlock, wrapped so the stepper sees a UNet-like module
    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.inp = torch.nn.Conv2d(8, 64, 3, padding=1)
            self.blk = psg.UNetBlock(64, 64, 128, 256, has_attention=True, num_heads=4)
            self.out = torch.nn.Conv2d(64, 8, 3, padding=1)
            self.te = psg.TimestepEmbedding(128)
            self.compute_dtype = torch.float32

        def forward(self, x, t, text):
            from pokemon_sprite_generator_amd import ops
            dt = torch.float32
            h = ops.conv2d(ops.nchw_to_nhwc(x, dt), self.inp.weight, self.inp.bias)
            pooled, tx = ops.text_pool(text, dt)
            h = self.blk.nhwc(h, self.te.embed(t, dt), pooled, tx)
            return ops.nhwc_to_nchw(ops.conv2d(h, self.out.weight, self.out.bias))
    return Tiny().cuda().eval()      # eval: dropout off so runs are comparable


def _batch():
    g = torch.Generator().manual_seed(7)
    return (torch.randn(4, 8, 9, 9, generator=g), torch.randn(4, 32, 256, generator=g),
            torch.tensor([10, 400, 700, 999]), torch.randn(4, 8, 9, 9, generator=g))


def _step(model, lat, txt, t, nz, distributed, bucket_bytes=1 << 16, bucket_dtype="fp32"):
    import pokemon_sprite_generator_amd as psg
    st = psg.DiffusionStepper(model, psg.NoiseScheduler(), lr=1e-3, weight_decay=0.0, max_grad_norm=1e9,
                              distributed=distributed, bucket_bytes=bucket_bytes,
                              grad_bucket_dtype=torch.bfloat16 if bucket_dtype == "bf16" else torch.float32)
    model_mode = model.training
    out = st.train_step(lat.cuda(), txt.cuda(), t.cuda(), nz.cuda())
    model.train(model_mode)
    torch.cuda.synchronize()
    return st, out


def _worker(rank, world, port, q, backend="gloo", bucket_dtype="fp32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    if backend == "nccl":                     # RCCL: one rank per GPU
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _make()
        if rank == 1:                          # replicas must be made equal by the stepper's broadcast, not by seeding
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.5)
        lat, txt, t, nz = _batch()
        sl = slice(2 * rank, 2 * rank + 2)
        # DiffusionStepper.train_step puts the model in train mode; keep dropout off by zeroing p via eval-time block
        import pokemon_sprite_generator_amd.unet as U
        U.ATTN_DROPOUT = 0.0
        st, out = _step(model, lat[sl], txt[sl], t[sl], nz[sl], True, bucket_dtype=bucket_dtype)
        assert st.reducer is not None and st.reducer.world == 2 and len(st.reducer.buckets) > 1
        assert st.reducer.avg_in_collective == (backend == "nccl")
        q.put((rank, float(out["loss"].item()), st.arena.flat.detach().cpu().numpy(), {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, None, traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def _cases():
    out = [("gloo", "fp32"), ("gloo", "bf16")]
    if torch.cuda.device_count() >= 2:         # RCCL needs one GPU per rank: runs on multi-GPU nodes, skipped on a 1-GPU box
        out += [("nccl", "fp32"), ("nccl", "bf16")]
    return out


def test_rccl_variant_is_collected_or_skipped():
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL 2-rank test needs >= 2 GPUs (one rank per GPU); this box has %d" % torch.cuda.device_count())


@pytest.mark.parametrize("backend,bucket_dtype", _cases())
def test_two_rank_step_matches_single_process(backend, bucket_dtype):
    import pokemon_sprite_generator_amd.unet as U
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend, bucket_dtype)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] is not None, r[2]
    res = [(r[0], r[1], torch.from_numpy(r[2]), {k: torch.from_numpy(v) for k, v in r[3].items()}) for r in res]
    # single process, whole batch
    old = U.ATTN_DROPOUT
    U.ATTN_DROPOUT = 0.0
    try:
        model = _make()
        lat, txt, t, nz = _batch()
        st, out = _step(model, lat, txt, t, nz, False)
    finally:
        U.ATTN_DROPOUT = old
    ref_flat = st.arena.flat.detach().cpu()
    # both ranks hold the same averaged gradients == the global-batch gradients
    assert torch.allclose(res[0][2], res[1][2], rtol=0, atol=0), "ranks disagree after all-reduce"
    err = float((res[0][2] - ref_flat).abs().max() / ref_flat.abs().max())
    bar = 1e-4 if bucket_dtype == "fp32" else 8e-3                # bf16 buckets round each rank's gradient to 8 bits
    assert err < bar, f"averaged shard gradients vs global-batch gradients: {err}"
    assert abs(0.5 * (res[0][1] + res[1][1]) - float(out["loss"].item())) < 1e-5
    # identical parameter update on every rank and equal to the single-process update
    sd = model.state_dict()
    for k in sd:
        assert torch.equal(res[0][3][k], res[1][3][k]), k
        if sd[k].dtype.is_floating_point:
            assert float((res[0][3][k] - sd[k].cpu()).abs().max()) < (1e-5 if bucket_dtype == "fp32" else 2.5e-3), k


# ---------------------------------------------------------------------------------------------------------------------
# configs[3] rehearsed at production size on ONE GPU: the full-width 640 M-parameter U-Net, bf16 compute, the production
# bucket size (64 MiB minimum -> 27 buckets of the 2.56 GB gradient arena), 2 ranks x batch 2 over gloo on cuda:0.
# ---------------------------------------------------------------------------------------------------------------------
N_PARAMS = 640_488_456


def _full_batch():
    g = torch.Generator().manual_seed(11)
    return (torch.randn(4, 8, 27, 27, generator=g) * 1.2, torch.randn(4, 32, 256, generator=g),
            torch.tensor([10, 400, 700, 999]), torch.randn(4, 8, 27, 27, generator=g))


def _full_step(distributed, sl, bucket_dtype):
    import pokemon_sprite_generator_amd as psg
    import pokemon_sprite_generator_amd.unet as U
    from pokemon_sprite_generator_amd import ops
    U.ATTN_DROPOUT = 0.0                     # dropout seeds mix in the rank: off, so 2 x 2 samples == 1 x 4 samples
    # (no split-K forward / data gradient here: its split count depends on the batch, and another fp32 summation order flips
    #  individual bf16 roundings of the activations - 1e-2 differences that are not what this test is about)
    ops._SPLITK = False
    torch.manual_seed(0)
    unet = psg.UNet(compute_dtype=torch.bfloat16).cuda()
    if distributed and torch.distributed.get_rank() == 1:       # replicas are made equal by the broadcast, not the seed
        with torch.no_grad():
            unet.init_conv.weight.add_(0.25)
    st = psg.DiffusionStepper(unet, psg.NoiseScheduler(), lr=1e-3, weight_decay=0.0, max_grad_norm=1e9, distributed=distributed,
                              grad_bucket_dtype=torch.bfloat16 if bucket_dtype == "bf16" else torch.float32)
    lat, txt, t, nz = _full_batch()
    out = st.train_step(lat[sl].cuda(), txt[sl].cuda(), t[sl].cuda(), nz[sl].cuda())
    torch.cuda.synchronize()
    return st, out


def _exact_digest(flat):
    """Two exact integer digests of a float tensor's bit patterns (equal digests <=> equal bits, for all practical purposes)."""
    bits = flat.view(torch.int32).to(torch.int64)
    return int(bits.sum().item()), int((bits * ((torch.arange(bits.numel(), device=bits.device) % 8191) + 1)).sum().item())


def _full_worker(rank, world, port, q, bucket_dtype, ref_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st, out = _full_step(True, slice(2 * rank, 2 * rank + 2), bucket_dtype)
        red = st.reducer
        ref = torch.from_numpy(np.load(ref_path)).cuda()
        got = st.arena.flat
        err_max = float((got - ref).abs().max() / ref.abs().max())
        # per bucket, so that a small-gradient bucket cannot hide behind a large one
        worst_bucket = max(float((got[s:e] - ref[s:e]).norm() / (ref[s:e].norm() + 1e-30)) for s, e, _ in red.buckets)
        q.put((rank, {"loss": float(out["loss"].item()), "flag": int(out["nan_flag"].item()), "err_max": err_max, "worst_bucket": worst_bucket,
                      "grad_digest": _exact_digest(got), "param_digest": _exact_digest(st.params.flat),
                      "buckets": len(red.buckets), "early": red.launched_early, "bytes": red.bytes_per_step, "world": red.world,
                      "numel": int(st.arena.numel), "steps": st.steps_done()}))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_dtype", ["fp32", "bf16"])
def test_full_width_two_rank_step(bucket_dtype, tmp_path):
    """BASELINE configs[3] on the box we have: production arena, production buckets, bucket-ready callbacks from the real
    backward, the all-reduce overlapping it, NaN-flag reduce, clip, AdamW - ranks equal, == one process on the global batch."""
    import numpy as np
    import pokemon_sprite_generator_amd.unet as U
    from pokemon_sprite_generator_amd import ops
    old, old_sk = U.ATTN_DROPOUT, ops._SPLITK
    try:
        st, out = _full_step(False, slice(0, 4), "fp32")
        ref_loss = float(out["loss"].item())
        ref_path = str(tmp_path / "ref_grads.npy")
        np.save(ref_path, st.arena.flat.detach().cpu().numpy())
        numel = int(st.arena.numel)
        st.close()
        del st, out
        torch.cuda.empty_cache()
    finally:
        U.ATTN_DROPOUT, ops._SPLITK = old, old_sk
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_full_worker, args=(r, 2, port, q, bucket_dtype, ref_path)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert isinstance(r[1], dict), r[1]
    a, b = res[0][1], res[1][1]
    assert a["world"] == b["world"] == 2 and a["flag"] == b["flag"] == 0 and a["steps"] == b["steps"] == 1
    # the whole 2.56 GB arena crosses the ranks once per step, in 64+ MiB buckets
    assert a["numel"] == numel and N_PARAMS <= numel <= N_PARAMS + 8 * 478
    assert a["bytes"] == (4 if bucket_dtype == "fp32" else 2) * numel
    # (a bucket closes with the parameter that takes it past 64 MiB, and the big 3x3 weights are 59 MB each: 27 buckets of 64-123 MiB)
    assert 20 <= a["buckets"] <= 40 and a["buckets"] == b["buckets"], a["buckets"]
    # overlap really happens: every bucket but (at most) the last left from a bucket-ready callback, before finish()
    assert a["early"] >= a["buckets"] - 1 and b["early"] >= b["buckets"] - 1, (a["early"], b["early"], a["buckets"])
    # ranks hold bit-identical averaged gradients and took the bit-identical parameter update
    assert a["grad_digest"] == b["grad_digest"], "ranks disagree after the all-reduce"
    assert a["param_digest"] == b["param_digest"], "ranks took different updates"
    # == single process on the global batch (shards are per-sample independent; only summation order / bucket rounding differ)
    bar_max, bar_bucket = (1e-4, 1e-3) if bucket_dtype == "fp32" else (8e-3, 8e-3)
    assert a["err_max"] < bar_max and b["err_max"] < bar_max, (a["err_max"], b["err_max"])
    assert a["worst_bucket"] < bar_bucket, a["worst_bucket"]
    assert abs(0.5 * (a["loss"] + b["loss"]) - ref_loss) < 2e-5 * max(1.0, abs(ref_loss))


def test_bench_rehearsal_two_ranks_one_gpu():
    """`python bench.py --gpus 2 --rehearse` through the real launch_ranks path (parent never touches the GPU, one worker per
    rank, rendezvous, rank-0 JSON relay): what the driver runs at N = 8, on the one GPU this box has."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["nccl_world_size"] == 2 and j["collective_backend"] == "gloo"
    assert j["config"]["global_batch"] == 2 * j["config"]["per_gpu_batch"] and j["config"]["parallelism"] == "dp2"
    assert j["allreduce_bytes_per_step"] >= 4 * N_PARAMS
    # the start-up measurement chose how the exchange runs (over gloo through the host the overlapped form usually loses);
    # whichever it chose is what the timed steps did (the overlap plumbing itself: test_full_width_two_rank_step)
    mode = j["allreduce_mode"]
    assert set(mode["tuned"]["ms_per_step"]) == {"overlap+reserve", "overlap+reserve-few-rounds", "overlap", "deferred"} and mode["tuned"]["chosen"] in mode["tuned"]["ms_per_step"]
    if mode["overlap"]:
        assert j["allreduce_buckets_launched_during_backward"] >= j["allreduce_buckets"] - 1
    else:
        assert j["allreduce_buckets_launched_during_backward"] == 0 and mode["tuned"]["chosen"] == "deferred"
    assert j["nan_flag"] == 0 and j["final_loss"] > 0 and j["value"] > 0
    assert "roofline" in j and "rehearsal" in j


# ---------------------------------------------------------------------------------------------------------------------
# RCCL itself, on the one GPU this pool has: a process group of ONE rank over backend "nccl".  Every collective of the
# data-parallel step really executes on the RCCL communicator - parameter broadcast, the ncclAvg decision (MIN all-reduce),
# the bucketed async all-reduce (ReduceOp.AVG on fp32 and bf16 buckets) launched from the bucket-ready callbacks on the
# side stream, the MAX reduce of the NaN flag - and, the exchange being the identity, the step must equal the
# non-distributed step bit for bit (bf16 buckets: the gradient rounded once to bf16).
# ---------------------------------------------------------------------------------------------------------------------
def _rccl_world1_worker(port, q, bucket_dtype, full):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import pokemon_sprite_generator_amd as psg
        import pokemon_sprite_generator_amd.unet as U
        bd = torch.bfloat16 if bucket_dtype == "bf16" else torch.float32

        def run(distributed):
            U._SeedStream.counter = 0               # TRAIN mode, dropout on: both runs draw the same per-site seeds (rank 0 both times)
            if full:
                torch.manual_seed(0)
                model = psg.UNet(compute_dtype=torch.bfloat16).cuda()
                lat, txt, t, nz = _full_batch()
                kw = dict(lr=1e-3, weight_decay=0.0, max_grad_norm=1e9)
            else:
                model = _make()
                lat, txt, t, nz = _batch()
                kw = dict(lr=1e-3, weight_decay=0.0, max_grad_norm=1e9, bucket_bytes=1 << 16)
            # (no CU reserve: with it the tile choosers plan backward for 224 CUs - other split counts, another fp32 summation
            #  order - and the comparison would not be bit for bit)
            st = psg.DiffusionStepper(model, psg.NoiseScheduler(), distributed=distributed, grad_bucket_dtype=bd, ddp_cu_reserve=0, **kw)
            out = st.train_step(lat.cuda(), txt.cuda(), t.cuda(), nz.cuda())
            torch.cuda.synchronize()
            return st, out

        st0, out0 = run(False)
        g0, p0, l0 = st0.arena.flat.clone(), st0.params.flat.clone(), float(out0["loss"].item())
        st0.close()
        del st0
        st1, out1 = run("force")
        red = st1.reducer
        g1, p1 = st1.arena.flat, st1.params.flat
        want = g0 if bucket_dtype == "fp32" else g0.bfloat16().float()
        g1c, p1c = g1.clone(), p1.clone()
        loss1, flag1, gnorm1 = float(out1["loss"].item()), int(out1["nan_flag"].item()), float(out1["grad_norm"].item())
        early, steps1 = red.launched_early, st1.steps_done()   # (of the step above: the measurement below runs more steps, on other settings)
        tuned = None
        restored = None
        if not full:                                 # the start-up measurement that picks the exchange mode runs real steps ...
            lat, txt, t, nz = _batch()
            red.cu_reserve = 32
            before = (st1.params.flat.clone(), st1.optimizer._m.clone(), st1.optimizer._v.clone(), st1.steps_done())
            tuned = st1.autotune_exchange(lat.cuda(), txt.cuda(), t.cuda(), trials=2)
            # ... and leaves the model, the moments and the step count exactly where they were
            restored = bool(torch.equal(st1.params.flat, before[0]) and torch.equal(st1.optimizer._m, before[1])
                            and torch.equal(st1.optimizer._v, before[2]) and st1.steps_done() == before[3])
        g1, p1 = g1c, p1c
        q.put({"ok": True, "tuned": tuned, "restored": restored, "backend": dist.get_backend(), "world": red.world, "active": red.active, "avg": red.avg_in_collective,
               "buckets": len(red.buckets), "early": early, "bytes": red.bytes_per_step, "numel": int(st1.arena.numel),
               "grads_equal": bool(torch.equal(g1, want)), "grad_err": float((g1 - want).abs().max()),
               "params_equal": bool(torch.equal(p1, p0)) if bucket_dtype == "fp32" else None,
               "loss0": l0, "loss1": loss1, "flag": flag1,
               "gnorm0": float(out0["grad_norm"].item()), "gnorm1": gnorm1, "steps": steps1})
    except Exception:  # noqa: BLE001
        import traceback
        q.put({"ok": False, "err": traceback.format_exc()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_dtype,full", [("fp32", False), ("bf16", False), ("fp32", True)], ids=["tiny-fp32", "tiny-bf16", "full-fp32"])
def test_rccl_world1_step_is_the_single_process_step(bucket_dtype, full):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(_free_port(), q, bucket_dtype, full))
    p.start()
    r = q.get(timeout=900)
    p.join(timeout=120)
    assert r["ok"], r.get("err")
    assert r["backend"] == "nccl" and r["world"] == 1 and r["active"] and r["avg"], r           # ncclAvg inside the collective
    assert r["buckets"] > 1 and r["early"] >= r["buckets"] - 1, r                                  # launched from the callbacks during backward
    assert r["bytes"] == (4 if bucket_dtype == "fp32" else 2) * r["numel"]
    if full:
        assert 20 <= r["buckets"] <= 40 and N_PARAMS <= r["numel"] <= N_PARAMS + 8 * 478
    assert r["flag"] == 0 and r["steps"] == 1 and r["loss0"] == r["loss1"]
    assert r["grads_equal"], r["grad_err"]
    if bucket_dtype == "fp32":
        assert r["params_equal"] and r["gnorm0"] == r["gnorm1"]
    if not full:
        assert r["restored"] is True
        tn = r["tuned"]
        assert tn["chosen"] in tn["ms_per_step"] and set(tn["ms_per_step"]) == {"overlap+reserve", "overlap+reserve-few-rounds", "overlap", "deferred"}, tn


def test_bench_force_ddp_runs_rccl_on_one_gpu():
    """`python bench.py --gpus 1 --force-ddp`: the driver's N = 1 command with the RCCL exchange switched on."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-ddp", "--no-cpu-baseline", "--no-secondary",
                        "--no-profile", "--batch", "8", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["nccl_world_size"] == 1 and j["collective_backend"] == "nccl" and "force_ddp" in j
    assert j["allreduce_bytes_per_step"] >= 4 * N_PARAMS and j["allreduce_avg_in_collective"] is True
    assert j["allreduce_buckets_launched_during_backward"] >= j["allreduce_buckets"] - 1
    assert j["nan_flag"] == 0 and j["final_loss"] > 0

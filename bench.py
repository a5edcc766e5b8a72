#!/usr/bin/env python3
"""bench.py — U-Net train steps/s (per-GPU batch 256, 27x27x8 latents, 32x256 text) on N MI355X.

Default (`--config train_bf16_bs256`, BASELINE.json configs[2] / [3]): one "step" = the reference's train-loop body
(improved_diffusion_trainer.py:363-413): clamp + add_noise -> U-Net forward (train mode, dropout on) -> SmoothL1 ->
backward -> [RCCL all-reduce of the 640 M gradients] -> global-norm clip -> AdamW, bf16 MFMA compute with fp32 master
weights / accumulation; weak scaling (256 samples per GPU), value = bs256-steps completed by all ranks per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--mode train|sample] [--batch B] [--dtype bf16|fp32]

--gpus N > 1: `python bench.py --gpus N` starts the N ranks itself (the parent never touches the GPU: it only starts one
child process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and relays rank 0's JSON line); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks already exist and each runs as a worker.
Other configs: `--config fwd_bwd_fp32_bs64` (configs[1]: forward + SmoothL1 + backward, fp32 exact MFMA, batch 64),
`--mode sample` (configs[4]: denoising steps of the 1000-step DDPM loop, 64 samples per GPU, hipGraph replay).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Forward FLOPs per sample, 2*MAC: FlopCounterMode on the reference in TRAIN mode (conv 59.10 + addmm 8.53 + mm 9.28 + bmm
# 0.56).  SURVEY.md §8(d)'s 70.83 was counted in eval / no_grad, where torch's fused MHA fast path hides the attention
# projections from the counter; the arithmetic is the same in both modes.
FWD_GFLOP_PER_SAMPLE = 77.46
FWD_BWD_GFLOP_PER_SAMPLE = 232.12
# Both constants come from ONE method - FlopCounterMode over the reference module with autograd recording (train or eval mode
# alike: 77.464 forward, 232.120 forward + SmoothL1 + backward; re-counted in round 4).  70.83 is what the same counter sees
# under no_grad, where nn.MultiheadAttention takes its fused fast path and the projections disappear from the count.
FLOP_MODEL = {"method": "torch.utils.flop_counter.FlopCounterMode on the reference U-Net, 2*MAC, autograd recording",
              "fwd_gflop_per_sample": 77.464, "fwd_bwd_gflop_per_sample": 232.120,
              "fwd_gflop_per_sample_no_grad_fastpath": 70.83,
              "note": "70.83 (SURVEY.md 8d as first written) hides the MHA projections behind torch's fused eval path; the fwd+bwd "
                      "figure was always the autograd-mode count, so step fractions of rounds 1-4 are comparable"}
PEAK_BF16 = 2.5e15                      # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32 = 157.3e12                     # fp32 matrix (= vector) rate
KINDS = ["conv_gemm(fwd gather)", "conv_gemm(dgrad gather)", "wgrad", "attention", "groupnorm"]


def _host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup quota (a GPU box hands
    out a CPU share; torch.get_num_threads() reports the whole machine and oversubscribes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:                    # noqa: BLE001
        pass
    return max(1, min(n, 64))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:                    # noqa: BLE001
        pass
    return "unknown"


def _oracle_weights():
    import torch
    import pokemon_sprite_generator_amd as psg
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in psg.UNet().state_dict().items()}
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, s in shapes.items():
        if k.endswith("emb_coeff"):
            sd[k] = torch.exp(torch.arange(s[0]) * -(torch.log(torch.tensor(10000.0)) / (s[0] - 1)))
        elif k.endswith("weight") and len(s) >= 2:
            fan = 1
            for d in s[1:]:
                fan *= d
            sd[k] = torch.randn(s, generator=g) * (fan ** -0.5)
        elif "norm" in k and k.endswith("weight"):
            sd[k] = torch.ones(s)
        else:
            sd[k] = torch.zeros(s)
    return sd, g


def cpu_baseline():
    """The oracle (CPU restatement of the reference path, pinned to the reference by tests/golden) timed on this host's
    cores on a bounded sample of the workload: configs[0] = one forward at B=1, t=500 (all cores and one
    thread), and the train step (forward + backward + clip + AdamW) at batch 2 on all cores and batch 1 on one thread."""
    import torch
    from oracle import unet_oracle as O
    sd, g = _oracle_weights()
    cores = _host_cores()
    tables = O.cosine_clipped_tables()

    def fwd_time(threads, reps):
        torch.set_num_threads(threads)
        x = torch.randn(1, 8, 27, 27, generator=g)
        text = torch.randn(1, 32, 256, generator=g)
        t = torch.tensor([500])
        best = 1e30
        with torch.no_grad():
            for _ in range(reps):
                t0 = time.time()
                O.unet_forward(sd, x, t, text, 8)
                best = min(best, time.time() - t0)
        return best

    def step_time(threads, B, reps):
        torch.set_num_threads(threads)
        x = torch.randn(B, 8, 27, 27, generator=g)
        text = torch.randn(B, 32, 256, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        noise = torch.randn(B, 8, 27, 27, generator=g)
        w = dict(sd)
        m = {k: torch.zeros_like(v) for k, v in w.items() if not k.endswith("emb_coeff")}
        v2 = {k: torch.zeros_like(v) for k, v in m.items()}
        best = 1e30
        for step in range(1, reps + 1):
            t0 = time.time()
            r = O.train_step_grads(w, x, text, t, noise, tables, 8)
            coef = O.clip_coef(r["grad_norm"], 1.0)
            for k, gk in r["grads"].items():
                w[k], m[k], v2[k] = O.adamw_update(w[k], gk * coef, m[k], v2[k], step, 1e-4, 0.9, 0.999, 1e-6, 0.01)
            best = min(best, time.time() - t0)
        return best

    f_all = fwd_time(cores, 3)
    f_one = fwd_time(1, 1)
    s_all = step_time(cores, 2, 2)
    s_one = step_time(1, 1, 1)
    torch.set_num_threads(cores)
    return {"value": (2 / s_all) / 256.0, "unit": "bs256-steps/s", "cores": cores, "kind": "port", "cpu": _cpu_model(),
            "sample": f"oracle train step (fwd+bwd+clip+AdamW), full-width U-Net, batch 2, fp32, dropout off, {cores} threads; best of 2 "
                      f"({s_all:.2f} s/step); scaled to batch-256 steps",
            "samples_per_s": 2 / s_all,
            "one_thread": {"train_step_s_batch1": s_one, "samples_per_s": 1 / s_one, "bs256_steps_per_s": (1 / s_one) / 256.0},
            "config0_forward_b1_t500": {"all_cores_s": f_all, "cores": cores, "one_thread_s": f_one,
                                        "what": "BASELINE configs[0]: single forward, B=1, t=500, 32x256 text, eval/no_grad"}}


def _build_id():
    """sha256[:12] of the kernel library being benched (ties committed PMC summaries to a build)."""
    from pokemon_sprite_generator_amd import _lib
    try:
        return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:12]
    except Exception:                    # noqa: BLE001
        return None


def _traffic_for(build_id, kernel, workload_key):
    """Memory-side bytes per launch of `kernel` from a committed rocprofv3 --pmc summary (tools/pmc_summary.py) of THIS
    build and workload; (None, reason) when no summary matches - PMC cannot be read from inside the process."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            tj = json.load(open(path))
        except Exception:                # noqa: BLE001
            continue
        if tj.get("build_id") != build_id or tj.get("workload") != workload_key:
            continue
        fam = tj.get("families", {}).get(kernel)
        if fam:
            return fam["traffic_bytes_per_launch"], os.path.relpath(path, ROOT) + ": " + tj.get("correction", "")
    return None, f"no profiles/*pmc_traffic*.json for build {build_id} / workload {workload_key}"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, rehearse=False):
    """Parent of a `python bench.py --gpus N` run: start one worker per GPU and relay rank 0's JSON line.  This process
    makes no HIP / torch.cuda call (a process that initialised the GPU must not exec or fork workers on this platform).
    Ranks > 0 write stdout + stderr to a scratch file each; when any rank fails its tail is relayed on stderr (a rank-1
    failure used to vanish into DEVNULL).  `rehearse`: every worker gets LOCAL_RANK 0 (one GPU, gloo transport)."""
    import tempfile
    port = _free_port()
    procs, logs = [], []
    tmp = tempfile.mkdtemp(prefix="psg_bench_")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if rehearse else r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PSG_BENCH_WORKER="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if r == 0:
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE))
            logs.append(None)
        else:
            f = open(os.path.join(tmp, f"rank{r}.log"), "wb")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=f, stderr=subprocess.STDOUT))
            logs.append(f)
    try:
        line, _ = procs[0].communicate(timeout=float(os.environ.get("PSG_BENCH_TIMEOUT", "1700")))
        codes = [procs[0].returncode] + [p.wait(timeout=120) for p in procs[1:]]
    except subprocess.TimeoutExpired:           # a rank hung (e.g. in the rendezvous): stop exactly the processes started here
        codes = []
        for p in procs:
            if p.poll() is None:
                p.kill()
                codes.append(124)
            else:
                codes.append(p.returncode)          # (a rank that had already died keeps its own exit code)
        sys.stderr.write("bench.py: a rank did not finish in time; workers killed\n")
        line = b""
    for r, f in enumerate(logs):
        if f is None:
            continue
        f.close()
        if any(codes):
            try:
                tail = open(f.name, "rb").read()[-4000:].decode(errors="replace")
            except OSError:
                tail = ""
            sys.stderr.write(f"---- bench.py rank {r} (exit {codes[r]}) ----\n{tail}\n")
        try:
            os.unlink(f.name)
        except OSError:
            pass
    try:
        os.rmdir(tmp)
    except OSError:
        pass
    sys.stdout.write(line.decode() if line else "")
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="train_bf16_bs256", choices=["train_bf16_bs256", "fwd_bwd_fp32_bs64"])
    ap.add_argument("--mode", default="train", choices=["train", "sample"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (train: 256 / 64 by config; sample: 64)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32"])
    ap.add_argument("--grad-bucket-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--no-graph", action="store_true", help="sample mode: eager loop instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event roofline leg")
    ap.add_argument("--profile-steps", type=int, default=3, help="steps of the exclusive-kernel roofline leg")
    ap.add_argument("--rehearse", action="store_true", help="multi-rank plumbing check on ONE GPU: the N workers all use cuda:0 and meet "
                                                            "over gloo (default batch 8, 2 steps); everything else - launch_ranks, the "
                                                            "rendezvous, the bucketed all-reduce of the full 640 M-gradient arena, the "
                                                            "max-over-ranks timing, the JSON line - is the production path")
    ap.add_argument("--force-ddp", action="store_true", help="--gpus 1 only: a process group of ONE rank over RCCL (backend nccl); the "
                                                             "bucketed all-reduce (ncclAvg, async work handles on the side stream, "
                                                             "launched from the bucket-ready callbacks during backward) and the NaN-flag "
                                                             "MAX reduce really execute - the exchange is the identity, its cost is not")
    ap.add_argument("--no-ddp-tune", action="store_true", help="data parallel: skip the start-up measurement that chooses between the overlapped "
                                                               "(with / without a CU reserve) and the deferred gradient exchange")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] / configs[4] legs of the default run")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous check only: the ranks meet over gloo on the CPU, all-reduce a 1 "
                                                           "and rank 0 prints the world size (no GPU; tests/test_bench_cpu.py)")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env != args.gpus:
        if os.environ.get("PSG_BENCH_WORKER"):
            raise SystemExit(f"worker started with WORLD_SIZE={world_env}, expected {args.gpus}")
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], rehearse=args.rehearse))
    run_worker(args)


def run_worker(args):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if args.gpus > 1 else 1
    if args.dry_run:
        if os.environ.get("PSG_BENCH_FAIL_RANK") == str(rank):          # (tests/test_bench_cpu.py: a rank that dies before the rendezvous)
            print(f"rank {rank}: injected failure", flush=True)
            sys.exit(3)
        one = torch.ones(1)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
            torch.distributed.all_reduce(one)
            torch.distributed.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": int(one.item())}), flush=True)
        return
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    nccl_world, backend = 1, None
    force_ddp = bool(args.force_ddp) and world == 1
    if force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:                                # one GPU, N processes: gloo moves the buckets through the host
            backend = "gloo"
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        else:
            backend = "nccl"                             # nccl == RCCL on ROCm
            from pokemon_sprite_generator_amd import ddp as _ddp
            _ddp.configure_rccl()                        # channel cap = the CU reserve the tile choosers plan around
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        probe = torch.ones(1, device=dev)
        torch.distributed.all_reduce(probe)              # a collective really ran: its sum IS the number of ranks
        nccl_world = int(round(float(probe.item())))

    import pokemon_sprite_generator_amd as psg
    from pokemon_sprite_generator_amd import _lib, ops
    lib = _lib.init(local_rank)
    sample_mode = args.mode == "sample"
    fwd_bwd_only = (not sample_mode) and args.config == "fwd_bwd_fp32_bs64"
    dtype_name = args.dtype or ("fp32" if fwd_bwd_only else "bf16")
    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    B = args.batch or (8 if args.rehearse else (64 if (sample_mode or fwd_bwd_only) else 256))
    steps = args.steps if args.steps is not None else (2 if args.rehearse else (50 if sample_mode else 20))
    warmup = args.warmup if args.warmup is not None else (1 if args.rehearse else (5 if sample_mode else 3))
    torch.manual_seed(1234)                      # (replicas are made equal by the stepper's rank-0 broadcast, not by this)
    unet = psg.UNet(latent_dim=8, text_dim=256, time_emb_dim=128, num_heads=8, compute_dtype=dtype).to(dev)
    stepper = psg.DiffusionStepper(unet, psg.NoiseScheduler(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-6,
                                   max_grad_norm=1.0, distributed=("force" if force_ddp else world > 1),
                                   grad_bucket_dtype=torch.bfloat16 if args.grad_bucket_dtype == "bf16" else torch.float32)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)          # every rank draws its own shard

    def barrier():
        if world > 1 or force_ddp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def make_steps(kind, batch, graph=True):
        """(one_step, profile_step, state) of a workload; `kind` in train / fwd_bwd / sample.  The U-Net must already be
        in the workload's compute dtype."""
        latents = torch.randn(batch, 8, 27, 27, device=dev, generator=gen) * 1.2     # clamp(-3,3) applies inside the step
        text = torch.randn(batch, 32, 256, device=dev, generator=gen)
        if kind == "sample":
            noise_fn = lambda i, shape: torch.randn(shape, device=dev, generator=gen)
            run = stepper.sampler(text, batch, fast_sampling=False, use_graph=graph, noise_fn=noise_fn)

            def prof_factory():
                return stepper.sampler(text, batch, fast_sampling=False, use_graph=False, noise_fn=noise_fn).step
            def sample_step():
                run.step()
                return None
            return sample_step, prof_factory, run
        if kind == "fwd_bwd":
            def one_step():
                t = torch.randint(0, 1000, (batch,), device=dev, generator=gen)
                noise = torch.randn(batch, 8, 27, 27, device=dev, generator=gen)
                unet.train()
                stepper.flag.zero_()
                noisy = stepper.noise_scheduler.add_noise(latents, noise, t, clamp=True, flag=stepper.flag)
                stepper.arena.zero()
                eps = unet(noisy, t, text)
                loss, dpred = stepper.smooth_l1(eps, noise)
                eps.backward(dpred)
                stepper.arena.finalize()
                if stepper.reducer is not None:
                    stepper.reducer.finish()
                return {"loss": loss, "nan_flag": stepper.flag}
            return one_step, (lambda: one_step), None

        def one_step():
            t = torch.randint(0, 1000, (batch,), device=dev, generator=gen)
            noise = torch.randn(batch, 8, 27, 27, device=dev, generator=gen)
            return stepper.train_step(latents, text, t, noise)
        return one_step, (lambda: one_step), None

    step_stats = {}

    def timed(one_step, nsteps, nwarm, tag="main"):
        """Wall clock over exactly `nsteps` steps between two barriers (the contract's figure) + one HIP event between
        consecutive steps on the launch stream: per-step device times, reported as median / min."""
        out = None
        for _ in range(nwarm):
            out = one_step()
        barrier()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nsteps + 1)]
        t0 = time.perf_counter()
        for i in range(nsteps):
            evs[i].record()
            out = one_step()
        evs[nsteps].record()
        barrier()
        el = time.perf_counter() - t0
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(nsteps))
        step_stats[tag] = {"median": per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2]),
                           "min": per[0], "max": per[-1], "n": nsteps, "how": "HIP events between consecutive steps on the launch stream"}
        tmax = torch.tensor([el], device=dev, dtype=torch.float64)
        if world > 1 or force_ddp:
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        return float(tmax.item()), out

    def families(prof_factory, nsteps, peak):
        """Roofline leg (after a timed region): per-launch HIP-event times are only meaningful when a kernel has the GPU to
        itself, and the timed region overlaps the weight-gradient GEMMs (second stream) with the data-gradient chain - so
        the same step runs `nsteps` more times with that overlap switched off (and, in sample mode, eagerly: event records
        inside a replayed graph would not bracket the replayed kernels).  Every rank runs them (they hold the collective)."""
        overlap, ops.SideStream.enabled = ops.SideStream.enabled, False
        pstep = prof_factory()
        pstep()
        barrier()
        if rank == 0:
            _lib.check(lib.psg_profile_begin(), "psg_profile_begin")
        for _ in range(nsteps):
            pstep()
        barrier()
        fam = None
        if rank == 0:
            n = len(KINDS)
            ms, work, cnt, byt = (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)()
            _lib.check(lib.psg_profile_end(ms, work, cnt, n), "psg_profile_end")
            _lib.check(lib.psg_profile_bytes(byt, n), "psg_profile_bytes")
            fam = []
            for i, name in enumerate(KINDS):
                if cnt[i] == 0:
                    continue
                is_bytes = name == "groupnorm"
                ach = work[i] / (ms[i] * 1e-3)
                fam.append({"kernel": name, "launches_per_step": cnt[i] / nsteps, "ms_per_step": ms[i] / nsteps,
                            "avg_launch_us": 1e3 * ms[i] / cnt[i], "bound": "hbm" if is_bytes else "mfma",
                            "achieved": ach / (1e9 if is_bytes else 1e12), "unit": "GB/s" if is_bytes else "TFLOP/s",
                            "peak": 8000.0 if is_bytes else peak / 1e12, "frac": ach / (8e12 if is_bytes else peak),
                            "algorithmic_bytes_per_launch": byt[i] / cnt[i]})
        ops.SideStream.enabled = overlap
        return fam

    def roofline_of(fam, build_id, workload_key, nsteps):
        dom = max(fam, key=lambda f: f["ms_per_step"])
        # memory-side bytes per launch come from a SEPARATE rocprofv3 --pmc pass of this same command (PMC cannot be
        # read from inside the process): attached only when a committed summary matches this build AND workload
        traffic, traffic_src = _traffic_for(build_id, dom["kernel"], workload_key)
        return {"bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"],
                "frac": dom["frac"], "traffic": traffic, "algorithmic_bytes": dom.get("algorithmic_bytes_per_launch"),
                "traffic_ratio": (traffic / dom["algorithmic_bytes_per_launch"]) if (traffic and dom.get("algorithmic_bytes_per_launch")) else None,
                "traffic_unit": "bytes/launch (memory-side requests incl. Infinity-Cache hits)",
                "traffic_source": traffic_src, "kernel": dom["kernel"],
                "avg_launch_us": dom["avg_launch_us"], "launches_per_step": dom["launches_per_step"],
                "ms_per_step": dom["ms_per_step"],
                "how": "algorithmic 2*M*N*K FLOPs summed over the family's launches / HIP-event time on the launch stream, "
                       f"{nsteps} steps run after the timed region with the wgrad side stream off (exclusive kernels)"}

    kind = "sample" if sample_mode else ("fwd_bwd" if fwd_bwd_only else "train")
    graph = not args.no_graph
    if sample_mode and graph and warmup < 2:
        warmup = 2                                       # step 0 is eager, step 1 captures
    one_step, prof_factory, run = make_steps(kind, B, graph)
    if stepper.reducer is not None and kind != "sample" and not args.no_ddp_tune:
        stepper.reducer.autotune(one_step, trials=2 if args.rehearse else 3)      # (untimed, before the warm-up: every rank takes part)
    elapsed, out = timed(one_step, steps, warmup)
    if sample_mode:
        loss, flag = None, int(not bool(torch.isfinite(run.x).all().item()))
    else:
        loss, flag = float(out["loss"].item()), int(out["nan_flag"].item())
    peak = PEAK_BF16 if dtype_name == "bf16" else PEAK_F32
    fam = families(prof_factory, args.profile_steps, peak) if not args.no_profile else None

    # ---- the other single-GPU BASELINE configs, same process, after the headline's timed region (N = 1 default run only) ----
    secondary = None
    if world == 1 and kind == "train" and dtype_name == "bf16" and not args.no_secondary and not args.rehearse and args.batch is None:
        secondary = {}
        build_id = _build_id()
        # configs[1]: forward + SmoothL1 + backward, exact-fp32 MFMA, batch 64
        unet.set_compute_dtype(torch.float32)
        s1, p1, _ = make_steps("fwd_bwd", 64)
        el1, o1 = timed(s1, 5, 2, "fwd_bwd_fp32_bs64")
        f1 = families(p1, 1, PEAK_F32) if not args.no_profile else None
        sps1 = 5 * 64 / el1
        secondary["fwd_bwd_fp32_bs64"] = {
            "metric": "unet_fwd_bwd_steps_per_sec_bs64", "value": 5 / el1, "unit": "bs64-steps/s", "steps": 5, "warmup": 2,
            "ms_per_step": 1e3 * el1 / 5, "dtype": "fp32", "samples_per_s": sps1,
            "step_ms": step_stats.get("fwd_bwd_fp32_bs64"),
            "step_mfma_frac_of_peak": sps1 * FWD_BWD_GFLOP_PER_SAMPLE * 1e9 / PEAK_F32, "final_loss": float(o1["loss"].item()),
            "config": {"workload": "BASELINE configs[1]: add_noise + U-Net forward + SmoothL1 + backward (no optimizer), batch 64, "
                                   "fp32 (exact fp32 MFMA), train mode (dropout 0.05)"}}
        if f1:
            secondary["fwd_bwd_fp32_bs64"]["roofline"] = roofline_of(f1, build_id, "fwd_bwd_fp32_bs64", 1)
        unet.set_compute_dtype(torch.bfloat16)
        # configs[4] per GPU: denoising steps of the 1000-step DDPM loop, 64 samples, hipGraph replay
        s4, p4, run4 = make_steps("sample", 64, True)
        el4, _ = timed(s4, 30, 3, "sample_bf16_n64")
        f4 = families(p4, 3, PEAK_BF16) if not args.no_profile else None
        secondary["sample_bf16_n64"] = {
            "metric": "ddpm_denoise_steps_per_sec_n64", "value": 30 / el4, "unit": "denoise-steps/s (64 samples per GPU each)",
            "steps": 30, "warmup": 3, "ms_per_step": 1e3 * el4 / 30, "dtype": "bf16", "samples_per_s": 30 * 64 / el4,
            "step_ms": step_stats.get("sample_bf16_n64"),
            "step_mfma_frac_of_peak": (30 * 64 / el4) * FWD_GFLOP_PER_SAMPLE * 1e9 / PEAK_BF16,
            "finite": bool(torch.isfinite(run4.x).all().item()),
            "config": {"workload": "BASELINE configs[4] per GPU: DDPM sampling loop (ddpm_sample :508-569), 1000-step schedule, 64 samples, "
                                   "U-Net forward + update per step, hipGraph replay, bf16 MFMA; 30 of the 1000 steps timed"}}
        if f4:
            secondary["sample_bf16_n64"]["roofline"] = roofline_of(f4, build_id, "sample_bf16_n64", 3)
        unet.train()

    if rank == 0:
        steps_per_s = world * steps / elapsed
        build_id = _build_id()
        if sample_mode:
            metric, unit = "ddpm_denoise_steps_per_sec_n64", "denoise-steps/s (64 samples per GPU each)"
            value, gflop = steps_per_s * (B / 64.0), FWD_GFLOP_PER_SAMPLE
            workload_key = f"sample_{dtype_name}_n{B}"
            workload = (f"DDPM sampling loop (ddpm_sample :508-569), 1000-step schedule, {B} samples per GPU, U-Net forward + update per "
                        f"step, {'hipGraph replay' if not args.no_graph else 'eager'}, {dtype_name} MFMA; {steps} of the 1000 steps timed")
        elif fwd_bwd_only:
            metric, unit = "unet_fwd_bwd_steps_per_sec_bs64", "bs64-steps/s"
            value, gflop = steps_per_s * (B / 64.0), FWD_BWD_GFLOP_PER_SAMPLE
            workload_key = f"fwd_bwd_{dtype_name}_bs{B}"
            workload = (f"add_noise + U-Net forward + SmoothL1 + backward (no optimizer), U-Net 640M params, per-GPU batch {B}, "
                        f"{dtype_name} (exact fp32 MFMA), train mode (dropout 0.05)")
        else:
            metric, unit = "unet_train_steps_per_sec_bs256", "bs256-steps/s"
            value, gflop = steps_per_s * (B / 256.0), FWD_BWD_GFLOP_PER_SAMPLE
            workload_key = f"train_{dtype_name}_bs{B}"
            workload = (f"full train_step (add_noise+fwd+SmoothL1+bwd+clip+AdamW), U-Net 640M params, per-GPU batch {B}, "
                        f"27x27x8 latents, 32x256 text, 8 heads, train mode (dropout 0.05), {dtype_name} MFMA / fp32 master+accum")
        samples_per_s = steps_per_s * B
        res = {
            "metric": metric, "value": value, "unit": unit,
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": workload, "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "samples_per_s": samples_per_s, "step_ms": step_stats.get("main"), "flop_model": FLOP_MODEL,
            "step_mfma_frac_of_peak": samples_per_s * gflop * 1e9 / (peak * world),
            "final_loss": loss, "nan_flag": flag, "build_id": build_id,
            "nccl_world_size": nccl_world, "collective_backend": backend,
            "allreduce_bytes_per_step": (stepper.reducer.bytes_per_step if stepper.reducer is not None and not sample_mode else 0),
        }
        if stepper.reducer is not None and not sample_mode:
            res["allreduce_buckets"] = len(stepper.reducer.buckets)
            res["allreduce_buckets_launched_during_backward"] = stepper.reducer.launched_early
            res["allreduce_avg_in_collective"] = stepper.reducer.avg_in_collective
            res["allreduce_mode"] = {"overlap": bool(stepper.reducer.overlap), "cu_reserve": stepper.reducer.cu_reserve,
                                     "reserve_rounds": stepper.reducer.reserve_rounds,
                                     "rccl_max_channels": os.environ.get("NCCL_MAX_NCHANNELS"), "tuned": stepper.reducer.tuned}
        if force_ddp:
            res["force_ddp"] = "process group of ONE rank over RCCL: every collective of the data-parallel step executes (identity exchange)"
        if args.rehearse:
            res["rehearsal"] = "N workers on ONE GPU over gloo: plumbing check, not a throughput figure"
        if fam:
            res["roofline"] = roofline_of(fam, build_id, workload_key, args.profile_steps)
            res["kernel_families"] = fam
        if secondary:
            res["secondary"] = secondary
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:                      # noqa: BLE001
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if world > 1 or force_ddp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

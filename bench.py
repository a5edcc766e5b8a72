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

FWD_GFLOP_PER_SAMPLE = 70.83           # FlopCounterMode on the reference, 2*MAC
FWD_BWD_GFLOP_PER_SAMPLE = 232.12
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


def launch_ranks(n, argv):
    """Parent of a `python bench.py --gpus N` run: start one worker per GPU and relay rank 0's JSON line.  This process
    makes no HIP / torch.cuda call (a process that initialised the GPU must not exec or fork workers on this platform)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PSG_BENCH_WORKER="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    try:
        line, _ = procs[0].communicate(timeout=float(os.environ.get("PSG_BENCH_TIMEOUT", "1700")))
        codes = [procs[0].returncode] + [p.wait(timeout=120) for p in procs[1:]]
    except subprocess.TimeoutExpired:           # a rank hung (e.g. in the rendezvous): stop exactly the processes started here
        for p in procs:
            if p.poll() is None:
                p.kill()
        sys.stderr.write("bench.py: a rank did not finish in time; workers killed\n")
        return 124
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
    ap.add_argument("--dry-run", action="store_true", help="rendezvous check only: the ranks meet over gloo on the CPU, all-reduce a 1 "
                                                           "and rank 0 prints the world size (no GPU; tests/test_bench_cpu.py)")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env != args.gpus:
        if os.environ.get("PSG_BENCH_WORKER"):
            raise SystemExit(f"worker started with WORLD_SIZE={world_env}, expected {args.gpus}")
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    run_worker(args)


def run_worker(args):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if args.gpus > 1 else 1
    if args.dry_run:
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
    nccl_world = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        probe = torch.ones(1, device=dev)
        torch.distributed.all_reduce(probe)            # an RCCL collective really ran: its sum IS the number of ranks
        nccl_world = int(round(float(probe.item())))

    import pokemon_sprite_generator_amd as psg
    from pokemon_sprite_generator_amd import _lib, ops
    lib = _lib.init(local_rank)
    sample_mode = args.mode == "sample"
    fwd_bwd_only = (not sample_mode) and args.config == "fwd_bwd_fp32_bs64"
    dtype_name = args.dtype or ("fp32" if fwd_bwd_only else "bf16")
    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    B = args.batch or (64 if (sample_mode or fwd_bwd_only) else 256)
    steps = args.steps if args.steps is not None else (50 if sample_mode else 10)
    warmup = args.warmup if args.warmup is not None else (5 if sample_mode else 3)
    torch.manual_seed(1234)                      # (replicas are made equal by the stepper's rank-0 broadcast, not by this)
    unet = psg.UNet(latent_dim=8, text_dim=256, time_emb_dim=128, num_heads=8, compute_dtype=dtype).to(dev)
    stepper = psg.DiffusionStepper(unet, psg.NoiseScheduler(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-6,
                                   max_grad_norm=1.0, distributed=(world > 1),
                                   grad_bucket_dtype=torch.bfloat16 if args.grad_bucket_dtype == "bf16" else torch.float32)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)          # every rank draws its own shard
    latents = torch.randn(B, 8, 27, 27, device=dev, generator=gen) * 1.2     # clamp(-3,3) applies inside the step
    text = torch.randn(B, 32, 256, device=dev, generator=gen)

    if sample_mode:
        graph = not args.no_graph
        run = stepper.sampler(text, B, fast_sampling=False, use_graph=graph,
                              noise_fn=lambda i, shape: torch.randn(shape, device=dev, generator=gen))
        if graph and warmup < 2:
            warmup = 2                                       # step 0 is eager, step 1 captures

        def one_step():
            run.step()
            return None
    elif fwd_bwd_only:
        def one_step():
            t = torch.randint(0, 1000, (B,), device=dev, generator=gen)
            noise = torch.randn(B, 8, 27, 27, device=dev, generator=gen)
            unet.train()
            stepper.flag.zero_()
            noisy = stepper.noise_scheduler.add_noise(latents, noise, t, clamp=True, flag=stepper.flag)
            stepper.arena.zero()
            eps = unet(noisy, t, text)
            loss, dpred = stepper.smooth_l1(eps, noise)
            eps.backward(dpred)
            stepper.arena.finalize()
            return {"loss": loss, "nan_flag": stepper.flag}
    else:
        def one_step():
            t = torch.randint(0, 1000, (B,), device=dev, generator=gen)
            noise = torch.randn(B, 8, 27, 27, device=dev, generator=gen)
            return stepper.train_step(latents, text, t, noise)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(warmup):
        out = one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if sample_mode:
        loss, flag = None, int(not bool(torch.isfinite(run.x).all().item()))
    else:
        loss, flag = float(out["loss"].item()), int(out["nan_flag"].item())
    # Roofline leg (rank 0, after the timed region): per-launch HIP-event times are only meaningful when a kernel has
    # the GPU to itself, and the timed region above overlaps the weight-gradient GEMMs (second stream) with the
    # data-gradient chain - so the same step runs `profile_steps` more times with that overlap switched off (and, in
    # sample mode, eagerly: event records inside a replayed graph would not bracket the replayed kernels).
    prof = None
    if not args.no_profile:                   # every rank runs the extra steps (they contain the collective)
        overlap, ops.SideStream.enabled = ops.SideStream.enabled, False
        if sample_mode:
            prun = stepper.sampler(text, B, fast_sampling=False, use_graph=False,
                                   noise_fn=lambda i, shape: torch.randn(shape, device=dev, generator=gen))
            pstep = prun.step
        else:
            pstep = one_step
        pstep()
        barrier()
        if rank == 0:
            _lib.check(lib.psg_profile_begin(), "psg_profile_begin")
        for _ in range(args.profile_steps):
            pstep()
        barrier()
        if rank == 0:
            n = len(KINDS)
            ms, work, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)()
            _lib.check(lib.psg_profile_end(ms, work, cnt, n), "psg_profile_end")
            prof = [(KINDS[i], ms[i], work[i], cnt[i]) for i in range(n)]
        ops.SideStream.enabled = overlap
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0:
        steps_per_s = world * steps / elapsed
        peak = PEAK_BF16 if dtype_name == "bf16" else PEAK_F32
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
            "samples_per_s": samples_per_s,
            "step_mfma_frac_of_peak": samples_per_s * gflop * 1e9 / (peak * world),
            "final_loss": loss, "nan_flag": flag, "build_id": build_id,
            "nccl_world_size": nccl_world,
            "allreduce_bytes_per_step": (stepper.reducer.bytes_per_step if stepper.reducer is not None and not sample_mode and not fwd_bwd_only else 0),
        }
        if prof:
            fam = []
            for name, ms_k, work_k, cnt_k in prof:
                if cnt_k == 0:
                    continue
                is_bytes = name == "groupnorm"
                ach = work_k / (ms_k * 1e-3)
                fam.append({"kernel": name, "launches_per_step": cnt_k / args.profile_steps, "ms_per_step": ms_k / args.profile_steps,
                            "avg_launch_us": 1e3 * ms_k / cnt_k, "bound": "hbm" if is_bytes else "mfma",
                            "achieved": ach / (1e9 if is_bytes else 1e12), "unit": "GB/s" if is_bytes else "TFLOP/s",
                            "peak": 8000.0 if is_bytes else peak / 1e12, "frac": ach / (8e12 if is_bytes else peak)})
            dom = max(fam, key=lambda f: f["ms_per_step"])
            # memory-side bytes per launch come from a SEPARATE rocprofv3 --pmc pass of this same command (PMC cannot be
            # read from inside the process): attached only when a committed summary matches this build AND workload
            traffic, traffic_src = _traffic_for(build_id, dom["kernel"], workload_key)
            res["roofline"] = {"bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"],
                               "frac": dom["frac"], "traffic": traffic, "traffic_unit": "bytes/launch (memory-side requests incl. Infinity-Cache hits)",
                               "traffic_source": traffic_src, "kernel": dom["kernel"],
                               "avg_launch_us": dom["avg_launch_us"], "launches_per_step": dom["launches_per_step"],
                               "ms_per_step": dom["ms_per_step"],
                               "how": "algorithmic 2*M*N*K FLOPs summed over the family's launches / HIP-event time on the launch stream, "
                                      f"{args.profile_steps} steps run after the timed region with the wgrad side stream off (exclusive kernels)"}
            res["kernel_families"] = fam
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:                      # noqa: BLE001
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

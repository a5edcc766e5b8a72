"""CPU suite: `python bench.py --gpus N` starts its own ranks (the driver runs exactly that command line); the parent only
spawns workers and relays rank 0's JSON line.  `--dry-run` makes the workers meet over gloo instead of touching a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "PSG_BENCH_WORKER"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_self_launches_its_ranks():
    out = _run("--gpus", "2", "--dry-run")
    assert out == {"dry_run": True, "n_gpus": 2, "ranks_seen": 2}


def test_bench_single_rank_runs_in_process():
    assert _run("--dry-run") == {"dry_run": True, "n_gpus": 1, "ranks_seen": 1}


def test_bench_under_torchrun_env_is_a_worker():
    """Under torch.distributed.run the ranks already exist: WORLD_SIZE == --gpus means 'I am a worker', no re-spawn."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=e,
                                      stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    js = [[l for l in o.splitlines() if l.startswith("{")] for o in outs]
    assert len(js[0]) == 1 and json.loads(js[0][0])["ranks_seen"] == 2 and js[1] == []      # only rank 0 prints the line


def test_bench_relays_a_failed_rank():
    """A rank > 0 that dies must be readable from the driver's tail: its output and exit code are relayed on stderr (it used to
    go to DEVNULL), the surviving rank is stopped at the timeout, and the launcher exits non-zero."""
    e = dict(os.environ, PSG_BENCH_FAIL_RANK="1", PSG_BENCH_TIMEOUT="25")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "PSG_BENCH_WORKER"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=300, env=e)
    assert r.returncode != 0
    assert "rank 1 (exit 3)" in r.stderr and "injected failure" in r.stderr, r.stderr[-2000:]

"""`python bench.py --gpus N` launched plainly must run N ranks (or refuse) -- never fall back to one GPU silently.
(reference fan-out: one worker process per env, furniture/env/base.py:55-80; here one process per GPU.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_plain_launch_with_two_gpus_spawns_two_ranks_over_the_launcher():
    # CPU stand-in for the data path (gloo all-reduce instead of the step + RCCL gather): same spawning code, same rendezvous
    r = _run("--gpus", "2", "--launcher-selftest")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["launcher_selftest"] and out["n_gpus"] == 2 and out["rank_sum"] == 1.0


def test_plain_launch_refuses_when_fewer_gpus_are_visible():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run("--gpus", str(have + 2), "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert "GPUs requested" in r.stderr and "visible" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no result line: nothing ran on fewer GPUs

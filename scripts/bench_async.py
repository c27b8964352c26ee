"""Exploration timing of the asynchronous env (furniture_amd/async_env.py) on the benchmark workload: every env performs exactly
K steps, batches are formed by predicted cost.  Not the headline bench (bench.py steps all envs in lockstep); one JSON line."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # C queue + E queues + torch's stream must map to distinct hardware queues
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from furniture_amd.async_env import FurnitureAsyncBatchEnv
from furniture_amd.envs import make_config

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
frac = float(os.environ.get("ASYNC_CHEAP_MIN", "0.5"))
ratio = float(os.environ.get("ASYNC_COST_RATIO", "1.6"))
neq = int(os.environ.get("ASYNC_EQ", "2"))
env = FurnitureAsyncBatchEnv("Sawyer", n, config=make_config(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825",
                                                             max_episode_steps=150, seed=123), cheap_min_fraction=frac, cost_ratio=ratio, n_expensive_queues=neq,
                             use_near_hint=os.environ.get("ASYNC_NEAR", "1") == "1")
env.reset()
dev = env.device
g = torch.Generator(device=dev); g.manual_seed(123)
steps = np.zeros(n, dtype=np.int64)


def run(k):
    target = steps.max() + k if steps.max() == steps.min() else None
    goal = steps + k
    ids = np.arange(n)
    env.send(torch.empty((n, env.dof), device=dev).uniform_(-1, 1, generator=g), ids)
    left = n * k
    while left:
        ids, obs, rew, done, info = env.recv()
        steps[ids] += 1
        left -= len(ids)
        nxt = ids[steps[ids] < goal[ids]]
        if len(nxt):
            env.send(torch.empty((len(nxt), env.dof), device=dev).uniform_(-1, 1, generator=g), nxt)


def run_for(seconds):
    """aggregate throughput: keep every env busy for a fixed wall time (no per-env step quota, as an asynchronous learner runs)"""
    ids = np.arange(n)
    env.send(torch.empty((n, env.dof), device=dev).uniform_(-1, 1, generator=g), ids)
    t_end = time.perf_counter() + seconds
    total = 0
    while time.perf_counter() < t_end:
        ids, obs, rew, done, info = env.recv()
        steps[ids] += 1
        total += len(ids)
        env.send(torch.empty((len(ids), env.dof), device=dev).uniform_(-1, 1, generator=g), ids)
    for ids, *_ in env.drain():
        steps[ids] += 1
        total += len(ids)
    return total


if os.environ.get("ASYNC_SECONDS"):
    run(3)
    torch.cuda.synchronize()
    s0 = steps.copy()
    t0 = time.perf_counter()
    total = run_for(float(os.environ["ASYNC_SECONDS"]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    d = steps - s0
    print(json.dumps({"workload": "async (aggregate, fixed wall time) FurnitureSawyerEnv + table_lack_0825", "envs": n, "seconds": dt,
                      "env_steps_per_s": total / dt, "steps_per_env_min_med_max": [int(d.min()), float(np.median(d)), int(d.max())],
                      "expensive_fraction": env.stats["expensive_envs"] / max(1, env.stats["expensive_envs"] + env.stats["cheap_envs"]),
                      "params": {"cheap_min": frac, "cost_ratio": ratio, "e_queues": neq, "near": os.environ.get("ASYNC_NEAR", "1")}}))
    env.close()
    sys.exit(0)

run(3)
torch.cuda.synchronize()
env.stats.update(cheap_batches=0, expensive_batches=0, cheap_envs=0, expensive_envs=0)
t0 = time.perf_counter()
run(K)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = env.stats
print(json.dumps({"workload": "async FurnitureSawyerEnv + table_lack_0825", "envs": n, "steps_per_env": K, "env_steps_per_s": n * K / dt,
                  "ms_per_round": dt / K * 1e3, "cheap_batches": st["cheap_batches"], "expensive_batches": st["expensive_batches"],
                  "mean_cheap_batch": st["cheap_envs"] / max(1, st["cheap_batches"]), "mean_expensive_batch": st["expensive_envs"] / max(1, st["expensive_batches"]),
                  "expensive_fraction": st["expensive_envs"] / max(1, st["expensive_envs"] + st["cheap_envs"]),
                  "params": {"cheap_min": frac, "cost_ratio": ratio, "e_queues": neq}, "obs_finite": bool(torch.isfinite(env.b._obs).all())}))
env.close()

"""Exploration timing of the scripted pick-and-attach scenario ("add a scripted-attach scenario to time and verify
A9 / A10"): every env of the batch runs furniture_amd.scripted.PickAndAttach under control_type ik_quaternion, so -- unlike the
random-action benchmark, where ~3 % of the envs touch a part -- the whole batch grips, carries and connects.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from furniture_amd.envs import FurnitureBatchEnv, make_config
from furniture_amd.scripted import FULL_TABLE, PickAndAttach

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
legs = FULL_TABLE if len(sys.argv) > 2 and sys.argv[2] == "full" else (0,)   # "full": all four legs, one episode
env = FurnitureBatchEnv("Sawyer", n, config=make_config(unity=False, record_vid=False, control_type="ik_quaternion", furniture_name="table_lack_0825",
                                                        max_episode_steps=20000, seed=123), auto_reset=False)
ob = env.reset()
pol = PickAndAttach(env.model, n)
stats = dict(steps=0, t_env=0.0)


def step(a):
    t0 = time.perf_counter()
    out = env.step(a)
    torch.cuda.synchronize()
    stats["t_env"] += time.perf_counter() - t0
    stats["steps"] += 1
    return out


t0 = time.perf_counter()
total, ncon, ob = pol.run(step, ob, legs=legs)
dt = time.perf_counter() - t0
print(json.dumps({"workload": "scripted assembly (furniture_amd.scripted), FurnitureSawyerEnv + table_lack_0825, ik_quaternion (150 substeps per step)", "envs": n,
                  "legs": list(legs),
                  "steps": stats["steps"], "connected_fraction": float((ncon == len(legs)).mean()),
                  "num_connected_histogram": np.bincount(ncon, minlength=len(legs) + 1).tolist(), "env_steps_per_s_device": n * stats["steps"] / stats["t_env"],
                  "ms_per_step_device": stats["t_env"] / stats["steps"] * 1e3, "physics_substeps_per_s": 150 * n * stats["steps"] / stats["t_env"],
                  "wall_s_incl_host_policy": dt, "mean_reward": float(total.mean())}))
env.close()

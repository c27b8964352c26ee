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
mode = sys.argv[2] if len(sys.argv) > 2 else "one"   # one: leg 0; full: all four legs, one episode; dense: the dense-reward env's recipe
legs, conns, hover, kw = (0,), None, 0.03, {}
if mode == "full":
    legs = FULL_TABLE
elif mode == "dense":
    from furniture_amd.dense import dense_subtasks
    from furniture_amd.envs import DENSE_OVERRIDES
    from furniture_amd.mjcf.model import load_compiled
    sub = dense_subtasks(load_compiled("Sawyer", "table_lack_0825"))[0]
    legs, conns, hover, kw = [int(d["leg_part"]) for d in sub], [int(d["k_table"]) for d in sub], 0.008, dict(DENSE_OVERRIDES)
kw.update(unity=False, record_vid=False, control_type="ik_quaternion", furniture_name="table_lack_0825", max_episode_steps=20000, seed=123)
env = FurnitureBatchEnv("Sawyer", n, config=make_config(**kw), auto_reset=False, dense=(mode == "dense"))
ob = env.reset()
pol = PickAndAttach(env.model, n, hover=hover)
done_any, succ = np.zeros(n, bool), np.zeros(n, bool)
stats = dict(steps=0, t_env=0.0)


def step(a):
    t0 = time.perf_counter()
    out = env.step(a)
    torch.cuda.synchronize()
    global done_any, succ
    d = out[2].cpu().numpy().astype(bool)
    succ |= ~done_any & d & (out[3]["episode_success"].cpu().numpy() != 0)
    if mode == "dense":   # the dense env ends the episode on a drop / wrong connection: nothing counts after the first done
        out = (out[0], torch.where(torch.as_tensor(done_any, device=out[1].device), torch.zeros_like(out[1]), out[1]), out[2], out[3])
    done_any |= d
    stats["t_env"] += time.perf_counter() - t0
    stats["steps"] += 1
    return out


t0 = time.perf_counter()
total, ncon, ob = pol.run(step, ob, legs=legs, table_connectors=conns)
dt = time.perf_counter() - t0
print(json.dumps({"workload": "scripted assembly (furniture_amd.scripted), FurnitureSawyerEnv + table_lack_0825, ik_quaternion (150 substeps per step)", "envs": n,
                  "mode": mode, "legs": list(legs), "episode_success_fraction": float(succ.mean()),
                  "steps": stats["steps"], "connected_fraction": float((ncon == len(legs)).mean()),
                  "num_connected_histogram": np.bincount(ncon, minlength=len(legs) + 1).tolist(), "env_steps_per_s_device": n * stats["steps"] / stats["t_env"],
                  "ms_per_step_device": stats["t_env"] / stats["steps"] * 1e3, "physics_substeps_per_s": 150 * n * stats["steps"] / stats["t_env"],
                  "wall_s_incl_host_policy": dt, "mean_reward": float(total.mean())}))
env.close()

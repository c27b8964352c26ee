"""Exploration timing of BASELINE config 5 (mixed-furniture batch) on one GPU: env-steps/s of FurnitureMixedBatchEnv.
Not the headline bench (bench.py is config 2); prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from furniture_amd.envs import make_config
from furniture_amd.mixed import FurnitureMixedBatchEnv

names = (sys.argv[1] if len(sys.argv) > 1 else "table_lack_0825,chair_agne_0007,shelf_ivar_0678").split(",")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096 // len(names) * len(names)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
env = FurnitureMixedBatchEnv("Sawyer", names, n, config=make_config(unity=False, record_vid=False, control_type="impedance", max_episode_steps=150))
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(123)
a = torch.empty((n, env.dof), device=env.device)
for _ in range(2):
    env.step(a.uniform_(-1, 1, generator=g))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    ob, rew, done, info = env.step(a.uniform_(-1, 1, generator=g))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"workload": "mixed batch " + "/".join(names), "envs": n, "steps": steps, "env_steps_per_s": n * steps / dt,
                  "ms_per_step": dt / steps * 1e3, "obs_dim_padded": env.obs_dim, "obs_finite": bool(torch.isfinite(env.obs).all())}))
env.close()

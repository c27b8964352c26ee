"""Development: find the slowest envs under deterministic actions on the GPU and replay one on the fp64 oracle to compare
Newton iteration counts (fp32 device vs fp64 oracle, same env, same actions)."""
import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
os.environ.setdefault("FSIM_LIB", os.path.join(R, "furniture_amd", "csrc", "libfsim_prof.so"))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
from oracle.oracle_sim import lib
from tests.scenarios import counter_actions
m = load_compiled("Sawyer", "table_lack_0825")
N, T = 1024, 8
cfg = default_config(); cfg.max_episode_steps = 150
sim = FSim(m, N, config=cfg)
sim.set_reset_tables(*ResetTableSampler(m, make_config(), 123, 0, N).draw())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev)
sim.reset(None, obs); sim.sync()
its = np.zeros((T, N))
for t in range(T):
    a = np.stack([counter_actions(123, i, t, 9) for i in range(N)])
    act.copy_(torch.as_tensor(a)); torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info); sim.sync()
    pall = sim.get_state("qacc")["qacc"].view(torch.int32)[:, :39].cpu().numpy().astype(np.int64)
    its[t] = pall[:, 6] / np.maximum(1, pall[:, 5])
worst = np.argsort(-its[-1])[:3]
print("device: newton it/substep of the 3 slowest envs at the last step:", [(int(e), round(float(its[-1, e]), 2)) for e in worst])
for e in worst[:2]:
    env = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + int(e)))
    env.reset()
    h = env.sim._h
    for tol in (1e-8,):
        env.sim.set_solver(100, tol, "newton")
    rows = []
    for t in range(T):
        cnt = []
        orig = env.sim.step
        def st():
            r = orig(); cnt.append(lib().osim_last_solver_iters(h)); return r
        env.sim.step = st
        env.step(counter_actions(123, int(e), t, 9).astype(np.float64))
        env.sim.step = orig
        rows.append(np.mean(cnt))
    print("env %d: device it/substep per step %s | oracle (fp64, tol 1e-8) %s" % (e, np.round(its[:, e], 2).tolist(), np.round(rows, 2).tolist()))

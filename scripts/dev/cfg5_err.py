import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, INFO_DIM, default_config
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
from tests.scenarios import counter_actions
for name in ["chair_agne_0007", "shelf_ivar_0678"]:
    m = load_compiled("Sawyer", name)
    n = 2
    cfg = default_config(); cfg.max_episode_steps = 150; cfg.auto_reset = 0
    sim = FSim(m, n, config=cfg)
    envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + i, solver_tolerance=1e-10)) for i in range(n)]
    obs_o = [e.reset() for e in envs]
    sim.set_reset_tables(np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]), np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs]))
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs); sim.sync()
    for e in range(n):
        d = np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(obs_o[e]))
        print(name, "reset env", e, "max err", d.max(), "at", d.argmax(), "of", len(d), "nparts", m.nparts)
    dof = sim.dof_action
    act = torch.zeros((n, dof), device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev); info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    for t in range(3):
        a = np.stack([counter_actions(321, i, t, dof) for i in range(n)])
        act.copy_(torch.as_tensor(a)); torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info); sim.sync()
        for e in range(n):
            ob, r, d_, _ = envs[e].step(a[e])
            d = np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(ob))
            print(name, "step", t, "env", e, "max err", d.max(), "at", d.argmax(), "rew", float(rew[e]), r)
    sim.close()

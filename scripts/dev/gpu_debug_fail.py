"""Find envs that go unstable on the GPU under random actions and replay them on the CPU oracle (development aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
m = load_compiled("Sawyer", "table_lack_0825")
N = 4096
cfg = default_config(); cfg.max_episode_steps = 150; cfg.auto_reset = 0
sim = FSim(m, N, config=cfg)
sampler = ResetTableSampler(m, make_config(), 123, 0, N)
sim.set_reset_tables(*sampler.draw())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev); g = torch.Generator(device=dev); g.manual_seed(123)
sim.reset(None, obs); sim.sync()
gn = m.meta["geom_names"]
found = 0
for t in range(10):
    pre = {k: v.cpu().numpy() for k, v in sim.get_state("qpos", "qvel", "qacc_warmstart", "qfrc_bias").items()}
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info); sim.sync()
    fails = np.where(info[:, 2].cpu().numpy() != 0)[0]
    print("step", t, "fails", len(fails))
    for e in fails[:3]:
        if found >= 6: break
        found += 1
        o = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, solver_tolerance=1e-10)); o.reset()
        o.sim.data.qpos[:] = pre["qpos"][e]; o.sim.data.qvel[:] = pre["qvel"][e]; o.sim.data.qacc_warmstart[:] = pre["qacc_warmstart"][e]; o.sim.data.qfrc_bias[:] = pre["qfrc_bias"][e]
        o.sim.data.xfrc_applied[:] = 0
        a = act[e].cpu().numpy().astype(np.float64)
        # replay substep by substep to see where/if the oracle blows up
        aa = a.copy(); aa[-2] = -1 if a[-2] < 0 else 1
        ctrl = o._setup_action(aa[:-1]); o.sim.data.ctrl[:] = ctrl
        status = "ok"; maxv = 0; maxit = 0
        for k in range(50):
            try:
                o.sim.step()
            except Exception as ex:
                status = "oracle unstable at substep %d: %s" % (k, ex); break
            maxv = max(maxv, np.abs(o.sim.data.qvel).max()); maxit = max(maxit, o.sim.last_solver_iters)
        cons = [(gn[a_], gn[b_]) for a_, b_ in o.sim.contacts() if a_ != m.floor_geomid[0]]
        print("  env", e, "action", np.round(a, 2), "|", status, "max|qvel| %.1f max newton it %d" % (maxv, maxit), "nonfloor contacts", cons[:6], "arm q", np.round(o.sim.data.qpos[:9], 3))

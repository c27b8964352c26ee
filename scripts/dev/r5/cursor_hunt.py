"""development: the Cursor agent on the device against the Python oracle env under random 15-dof actions with frequent select
requests, at a scale the suite does not run (tests/test_gpu_parity.py: 2 envs x 6 steps + a scripted attach): per step the envs whose
observation / integer words agree.  usage: cursor_hunt.py <furniture> <n> <steps>"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, INFO_DIM, default_config
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig

furn, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = load_compiled("Cursor", furn)
cfg = default_config()
cfg.max_episode_steps, cfg.auto_reset = 1000, 0
sim = FSim(m, n, config=cfg)
envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=1000, seed=200 + i)) for i in range(n)]
obs_o = [e.reset() for e in envs]
sim.set_reset_tables(np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]), None)
dev = sim.device
obs = torch.zeros((n, sim.obs_dim), device=dev)
sim.reset(None, obs)
sim.sync()
print("reset max %.2e" % max(np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(obs_o[e])).max() for e in range(n)))
act, rew = torch.zeros((n, 15), device=dev), torch.zeros(n, device=dev)
done, info = torch.zeros(n, dtype=torch.uint8, device=dev), torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
rng = np.random.RandomState(1)
alive = np.ones(n, dtype=bool)  # envs that have agreed so far (a discrete event taken differently ends the comparison of that env)
for t in range(steps):
    a = rng.uniform(-1, 1, (n, 15)).astype(np.float32)
    a[:, 6] = np.abs(a[:, 6]) * np.where(rng.rand(n) < 0.8, 1, -1)   # select mostly on
    a[:, 13] = np.abs(a[:, 13]) * np.where(rng.rand(n) < 0.8, 1, -1)
    # cursors drift towards the parts (they start 0.2 m to the sides): bias the moves to the origin half of the time
    act.copy_(torch.as_tensor(a))
    torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info)
    sim.sync()
    og, gi = obs.cpu().numpy(), info.cpu().numpy()
    cur = sim.get_state("cursor")["cursor"].cpu().numpy()
    nsel_d = nsel_o = 0
    bad = []
    for e in range(n):
        ob, r, d, inf = envs[e].step(a[e].astype(np.float64))
        if not alive[e]:
            continue
        dd = np.abs(og[e] - envs[e].flat_obs(ob)).max()
        sel_o = [(-1 if s is None else s) for s in envs[e]._cursor_selected]
        sel_d = [int(cur[e, 6]) - 1, int(cur[e, 7]) - 1]
        nsel_d += sum(s >= 0 for s in sel_d); nsel_o += sum(s >= 0 for s in sel_o)
        ok = dd < 2e-3 and sel_o == sel_d and gi[e, 0] == inf["num_connected"] and abs(float(rew[e]) - r) < 1e-4 and bool(done[e]) == d
        if not ok:
            bad.append((e, "%.1e" % dd, sel_d, sel_o, int(gi[e, 0]), inf["num_connected"]))
            alive[e] = False
    for e in [int(x) for x in os.environ.get("WATCH", "").split(",") if x]:
        names = m.meta["geom_names"]
        sim.physics_forward()
        cgd = sim.get_state("contact_geoms")["contact_geoms"][e].cpu().numpy().reshape(-1, 2)
        cd = sorted(set((names[a_], names[b_]) for a_, b_ in cgd if a_ >= 0 and ("cursor" in names[a_] or "cursor" in names[b_])))
        co = sorted(set((names[a_], names[b_]) for a_, b_ in envs[e].sim.contacts() if "cursor" in names[a_] or "cursor" in names[b_]))
        print("   env %d t %d act sel %+.2f %+.2f  dev sel %s cur %s | oracle sel %s cur %s" % (e, t, a[e, 6], a[e, 13], [int(cur[e, 6]) - 1, int(cur[e, 7]) - 1], np.round(cur[e, :6], 4).tolist(),
              [(-1 if s_ is None else s_) for s_ in envs[e]._cursor_selected], np.round(np.concatenate([envs[e]._cursor_pos(0), envs[e]._cursor_pos(1)]), 4).tolist()))
        stq = sim.get_state("qpos", "env_block")
        dq = np.abs(stq["qpos"][e].cpu().numpy() - envs[e].sim.data.qpos)
        print("      connect %+.2f  connect_step dev %d oracle %d  qpos diff per part %s" % (a[e, 14], int(stq["env_block"][e, 2]), envs[e]._connect_step, [float("%.1e" % dq[m.part_qposadr[i]:m.part_qposadr[i] + 7].max()) for i in range(m.nparts)]))
        print("      dev info fail %d overflow %d num_connected %d done %d | oracle fail %s  ncon dev %d" % (gi[e, 2], gi[e, 12], gi[e, 0], int(done[e]), envs[e]._fail, int((cgd[:, 0] >= 0).sum())))
        print("      cursor contacts dev %s" % cd)
        print("      cursor contacts ora %s" % co)
    print("t %2d  agreeing envs %3d / %d  selections held dev %d oracle %d  newly apart: %s" % (t, alive.sum(), n, nsel_d, nsel_o, bad[:4]))
sim.close()

import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config
from oracle.oracle_sim import OracleSim
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
cases = []
for t in range(5):
    pre = {k: v.cpu().numpy() for k, v in sim.get_state("qpos", "qvel", "qacc_warmstart", "qfrc_bias").items()}
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info); sim.sync()
    for e in np.where(info[:, 2].cpu().numpy() != 0)[0][:3]:
        cases.append((t, e, {k: v[e].copy() for k, v in pre.items()}, act[e].cpu().numpy().astype(np.float64)))
print("cases", [(c[0], c[1]) for c in cases])
one = FSim(m, 1, config=cfg)
rd = np.concatenate([m.arm_dofadr, m.grip_dofadr])
for (t, e, pre, a) in cases[:4]:
    aa = np.clip(a.copy(), -1, 1); aa[7] = -1 if a[7] < 0 else 1
    ctrl = m.ctrl_bias + m.ctrl_weight * np.concatenate([aa[:7], [aa[7], -aa[7]]])
    app = np.zeros(m.nv); app[rd] = pre["qfrc_bias"][rd]
    one.set_state(qpos=pre["qpos"][None], qvel=pre["qvel"][None], qacc_warmstart=pre["qacc_warmstart"][None], ctrl=ctrl[None], qfrc_applied=app[None], xfrc_applied=np.zeros((1, 30)))
    o = OracleSim(m); o.set_solver(100, 1e-10, "newton"); o.reset()
    o.data.qpos[:] = pre["qpos"]; o.data.qvel[:] = pre["qvel"]; o.data.qacc_warmstart[:] = pre["qacc_warmstart"]; o.data.ctrl[:] = ctrl; o.data.qfrc_applied[:] = app
    print("case step %d env %d" % (t, e))
    for k in range(50):
        one.physics_step(1)
        s = one.get_state("qacc", "qpos", "qvel", "solver_iters", "ncon", "contact_geoms")
        o.step()
        qa = s["qacc"][0].cpu().numpy(); dq = np.abs(s["qpos"][0].cpu().numpy() - o.data.qpos).max()
        da = np.abs(qa - o.data.qacc)
        bad = not np.isfinite(qa).all() or da.max() > 1e-2 * (1 + np.abs(o.data.qacc).max())
        if bad or k % 10 == 0:
            cg = s["contact_geoms"][0].cpu().numpy().reshape(-1, 2); cg = [(m.meta["geom_names"][x], m.meta["geom_names"][y]) for x, y in cg if x >= 0 and x != m.floor_geomid[0]]
            print("  sub %2d it gpu/orc %d/%d ncon %d/%d  max|qacc| gpu %.3g orc %.3g  dqacc %.3g at dof %d  dqpos %.2e nonfloor %s" % (k, int(s["solver_iters"][0]), o.last_solver_iters, int(s["ncon"][0]), o.ncon, float(np.nan_to_num(np.abs(qa), nan=1e30).max()), np.abs(o.data.qacc).max(), float(np.nan_to_num(da, nan=1e30).max()), int(np.nan_to_num(da, nan=1e30).argmax()), dq, cg[:4]))
        if k in (8, 9, 10) and t == cases[0][0] and e == cases[0][1]:
            gnm = m.meta["geom_names"]
            for g1, g2 in o.contacts():
                if g1 != m.floor_geomid[0]:
                    print("     orc contact", gnm[g1], gnm[g2], "pos1", np.round(o.data.geom_xpos[g1], 3), "pos2", np.round(o.data.geom_xpos[g2], 3), "z2", np.round(o.data.geom_xmat[g2].reshape(3, 3)[:, 2], 2))
            print("     arm q", np.round(o.data.qpos[:9], 3), "qvel", np.round(o.data.qvel[:9], 2))
        if bad and (not np.isfinite(qa).all()): break

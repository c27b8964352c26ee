"""How many envs would a state-based "robot hand near a part" class select, and how well does it predict the expensive envs?
(development aid for the multi-wave kernel: the class must be a function of the state, not of timing)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config

m = load_compiled("Sawyer", "table_lack_0825")
N = 2048
cfg = default_config(); cfg.max_episode_steps = 150
sim = FSim(m, N, config=cfg)
sampler = ResetTableSampler(m, make_config(), 123, 0, N)
sim.set_reset_tables(*sampler.draw())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev); g = torch.Generator(device=dev); g.manual_seed(123)
sim.reset(None, obs); sim.sync()
def arr(k):
    return np.asarray(m.arrays[k])
nparts = int(sim.obs_dim - 29) // 7
cg_body, cg_pos, cg_rb, ispart = arr("cg_body"), arr("cg_pos").reshape(-1, 3), arr("cg_rbound"), arr("cg_ispartcol")
part_rbody = arr("part_rbody")
def quat_rot(q, v):  # q wxyz [n,4], v [3]
    w, u = q[:, :1], q[:, 1:]
    t = 2 * np.cross(u, v[None, :]); return v[None, :] + w * t + np.cross(u, t)
for t in range(60):
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info); sim.sync()
    o = obs.cpu().numpy()
    eef = o[:, 7 * nparts + 16: 7 * nparts + 19]
    dmin = np.full(N, 1e9)
    for gi in np.nonzero(ispart)[0]:
        p = int(np.nonzero(part_rbody == cg_body[gi])[0][0])
        ctr = o[:, 7 * p: 7 * p + 3] + quat_rot(o[:, 7 * p + 3: 7 * p + 7], cg_pos[gi])
        dmin = np.minimum(dmin, np.linalg.norm(ctr - eef, axis=1) - cg_rb[gi])
    it = sim.get_state("solver_iters")["solver_iters"].cpu().numpy()
    if t % 5 == 4:
        slow = it >= 3
        print("step %2d: near(eef) <5cm %.3f <10cm %.3f <15cm %.3f <20cm %.3f | last-substep niter>=3: %.3f ; of those near<10cm %.2f near<15cm %.2f near<20cm %.2f" % (
            t, (dmin < .05).mean(), (dmin < .10).mean(), (dmin < .15).mean(), (dmin < .20).mean(), slow.mean(),
            (dmin[slow] < .10).mean() if slow.any() else 0, (dmin[slow] < .15).mean() if slow.any() else 0, (dmin[slow] < .20).mean() if slow.any() else 0))

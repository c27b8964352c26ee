"""development: what the slowest env of every step looks like (profile build): cost, Newton iterations, big-island sizes"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("FSIM_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "furniture_amd", "csrc", "libfsim_prof.so"))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config
m = load_compiled("Sawyer", "table_lack_0825")
N = 2048
cfg = default_config(); cfg.max_episode_steps = 150
sim = FSim(m, N, config=cfg)
sim.set_reset_tables(*ResetTableSampler(m, make_config(), 123, 0, N).draw())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev); g = torch.Generator(device=dev); g.manual_seed(123)
sim.reset(None, obs); sim.sync()
rows = []
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 80):
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    t0 = time.time(); sim.step(act, obs, rew, done, info); sim.sync(); dt = time.time() - t0
    p = sim.get_state("qacc")["qacc"].view(torch.int32).cpu().numpy().astype(np.int64)
    tot = (p[:, 1] + p[:, 3] + p[:, 4] + p[:, 16:22].sum(axis=1)) * 16
    e = int(np.argmax(tot))
    nsub, nit, maxit = p[e, 5], p[e, 6], p[e, 10]
    nbig, szsum, nover = p[e, 36] & 0x3ff, p[e, 35], (p[e, 36] >> 10) & 0x3f
    unc = (p[e, 36] >> 16) & 0xff
    rows.append((dt * 1e3, tot[e] / 1e6, nit / max(1, nsub), maxit, nbig, szsum / max(1, nbig), unc, np.sort(tot)[-5] / 1e6, nover))
    if t >= 8:
        print("step %2d: %.1f ms | worst env %4d: %.1f Mcyc, it/substep %.2f (max %d), solves with a big island %d (mean size %.1f), uncached-pair solves %d | 5th worst %.1f Mcyc | solves with an island > 31 dofs (at least, 6-bit counter) %d" % ((t, rows[-1][0], e) + rows[-1][1:]))
R = np.array(rows[8:])
print("mean step %.2f ms; worst env mean %.1f Mcyc; steps with worst > 16 Mcyc: %d of %d; mean big-island size of the worst env %.1f" % (R[:, 0].mean(), R[:, 1].mean(), (R[:, 1] > 16).sum(), len(R), R[:, 5].mean()))

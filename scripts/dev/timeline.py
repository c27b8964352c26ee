"""development: residency timeline of k_env_step from the -DFSIM_PROFILE -DFSIM_TIMELINE build (start / end tick and HW_ID per env)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config
m = load_compiled("Sawyer", "table_lack_0825")
N = int(os.environ.get("FSIM_PROF_N", "4096"))
cfg = default_config(); cfg.max_episode_steps = 150
sim = FSim(m, N, config=cfg)
sim.set_reset_tables(*ResetTableSampler(m, make_config(), 123, 0, N).draw())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev); g = torch.Generator(device=dev); g.manual_seed(123)
sim.reset(None, obs); sim.sync()
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    t0 = time.time(); sim.step(act, obs, rew, done, info); sim.sync(); dt = time.time() - t0
    p = sim.get_state("qacc")["qacc"].view(torch.int32).cpu().numpy().astype(np.int64)
    st, en, hw, xcc = p[:, 37], p[:, 38], p[:, 34], p[:, 33]  # (10 ns ticks of the device-wide 100 MHz counter)
    en = np.where(en < st, en + (1 << 31), en)
    t_0 = st.min(); st -= t_0; en -= t_0
    span = en.max()
    ev = np.concatenate([np.stack([st, np.ones(N)], 1), np.stack([en, -np.ones(N)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1])
    dtk = np.diff(ev[:, 0], append=ev[-1, 0])
    avg = (conc * dtk).sum() / max(1, span)
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
    key = xcc * 4096 + se * 64 + sh * 16 + cu
    # waves resident on one CU at the time the 300th workgroup starts
    tq = np.sort(st)[min(N - 1, 3000)]
    live = (st <= tq) & (en > tq)
    per_cu = np.bincount(np.unique(key[live], return_inverse=True)[1])
    print("   per env Mticks: entry->substeps %.3f | 50 substeps %.3f | after substeps (scan, reward, obs) %.3f | sum of the forward phases %.3f | integrate %.3f | guard %.3f" % (
        p[:, 32].mean() * 16 / 1e6, p[:, 35].mean() * 16 / 1e6, p[:, 36].mean() * 16 / 1e6, (p[:, 1] + p[:, 3] + p[:, 4]).mean() * 16 / 1e6 + p[:, 16:22].sum(axis=1).mean() * 16 / 1e6, p[:, 0].mean() * 16 / 1e6, p[:, 2].mean() * 16 / 1e6))
    print("step %d: wall %.2f ms | span %.2f Mticks | env duration mean %.2f max %.2f Mticks | concurrency max %d mean %.0f | distinct CUs %d | live waves per CU at mid-launch: min %d median %d max %d | first-start spread %.2f Mticks" % (
        t, dt * 1e3, span / 1e6, (en - st).mean() / 1e6, (en - st).max() / 1e6, conc.max(), avg, len(np.unique(key)), per_cu.min(), np.median(per_cu), per_cu.max(), np.sort(st)[min(N - 1, 2047)] / 1e6))

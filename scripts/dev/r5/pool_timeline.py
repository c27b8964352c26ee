"""development (round 5): where a pooled slab-step's time goes.  G slabs attached to one work pool (POOL=1) or on launches of their own
(POOL=0), one host thread per slab, T steps; afterwards the per-env start / end ticks of every slab's LAST step (the -DFSIM_TIMELINE
build leaves them in the debug rows) and the host-side post -> completion time of that step.
usage: FSIM_LIB=.../libfsim_tl.so G=8 POOL=1 python scripts/dev/r5/pool_timeline.py [steps]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ.setdefault("FSIM_LIB", os.path.join(ROOT, "furniture_amd", "csrc", "libfsim_tl.so"))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, FSimPool, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config

m = load_compiled("Sawyer", "table_lack_0825")
G = int(os.environ.get("G", "8")); N = int(os.environ.get("N", "4096")); ng = N // G
POOL = os.environ.get("POOL", "1") == "1"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = default_config(); cfg.max_episode_steps = 150


class Slab:
    pass


slabs = []
for g in range(G):
    sl = Slab(); sl.g = g
    sl.sim = FSim(m, ng, config=cfg)
    sl.sim.set_reset_tables(*ResetTableSampler(m, make_config(), 123, g * ng, ng).draw())
    dev = sl.sim.device
    sl.obs = torch.zeros((ng, sl.sim.obs_dim), device=dev); sl.rew = torch.zeros(ng, device=dev); sl.done = torch.zeros(ng, dtype=torch.uint8, device=dev)
    sl.info = torch.zeros((ng, INFO_DIM), dtype=torch.int32, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(123 + g)
    sl.act = torch.empty((T, ng, 9), device=dev).uniform_(-1, 1, generator=gen)
    sl.sim.reset(None, sl.obs); sl.sim.sync()
    sl.lat = []
    slabs.append(sl)
torch.cuda.synchronize()
pool = None
if POOL:
    pool = FSimPool()
    for sl in slabs:
        pool.attach(sl.sim)


def loop(sl):
    for t in range(T):
        t0 = time.perf_counter()
        sl.sim.step(sl.act[t], sl.obs, sl.rew, sl.done, sl.info)
        sl.sim.sync()
        sl.lat.append(time.perf_counter() - t0)


t_start = time.perf_counter()
th = [threading.Thread(target=loop, args=(sl,)) for sl in slabs]
[t.start() for t in th]
[t.join() for t in th]
wall = time.perf_counter() - t_start
print("G %d x %d envs, pool %d, %d steps: %.0f env-steps/s, slab-step latency mean %.2f ms (last 10 steps: %.2f)" % (
    G, ng, POOL, T, N * T / wall, np.mean([np.mean(sl.lat[5:]) for sl in slabs]) * 1e3, np.mean([np.mean(sl.lat[-10:]) for sl in slabs]) * 1e3))
MS = 1e5
allst, allen = [], []
for sl in slabs:
    p = sl.sim.get_state("qacc")["qacc"].view(torch.int32)[:, :48].cpu().numpy().astype(np.int64)
    eb = np.ascontiguousarray(sl.sim.get_state("env_block")["env_block"].cpu().numpy()).view(np.int32)
    st, en = p[:, 37], p[:, 38]
    en = np.where(en < st, en + (1 << 31), en)
    dur = en - st
    nit = eb[:, 35]
    k0 = st.min()
    print("slab %2d last step: host latency %.2f ms | device span %.2f ms | env duration mean %.2f p50 %.2f p90 %.2f max %.2f ms | 50-it envs mean %.2f ms | starts: 50 / 90 / 100 %% by +%.2f / %.2f / %.2f ms | longest env started +%.2f, %d it" % (
        sl.g, sl.lat[-1] * 1e3, (en.max() - k0) / MS, dur.mean() / MS, np.percentile(dur, 50) / MS, np.percentile(dur, 90) / MS, dur.max() / MS,
        dur[nit <= 51].mean() / MS if (nit <= 51).any() else -1, np.percentile(st - k0, 50) / MS, np.percentile(st - k0, 90) / MS, (st - k0).max() / MS,
        (st[np.argmax(dur)] - k0) / MS, nit[np.argmax(dur)]))
if pool:
    print(pool.stats())
    pool.close()

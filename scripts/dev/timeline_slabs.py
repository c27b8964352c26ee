"""development: where does a pipelined slab-step's time go?  G slabs stepped round-robin as bench.py does, on the
-DFSIM_PROFILE -DFSIM_TIMELINE build (start / end tick of every env); per slab-step: kernel span, the env that ended last (when it
started, how long it ran, whether the scheduler had it among the multi-wave envs), and the number of envs resident over time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("FSIM_LIB", os.path.join(ROOT, "furniture_amd", "csrc", "libfsim_tl.so"))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config

m = load_compiled("Sawyer", "table_lack_0825")
G = int(os.environ.get("G", "4")); N = int(os.environ.get("N", "4096")); ng = N // G
T = int(sys.argv[1]) if len(sys.argv) > 1 else 14
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
    sl.t = 0; sl.inflight = False; sl.rows = []; sl.prev_nit = None
    slabs.append(sl)
torch.cuda.synchronize()
K = int(os.environ.get("FSIM_MW_K", "150"))


def wait(sl):
    if not sl.inflight:
        return
    sl.sim.sync(); sl.inflight = False
    p = sl.sim.get_state("qacc")["qacc"].view(torch.int32)[:, :48].cpu().numpy().astype(np.int64)
    eb = np.ascontiguousarray(sl.sim.get_state("env_block")["env_block"].cpu().numpy()).view(np.int32)
    sl.rows.append((sl.t - 1, p[:, 37], p[:, 38], sl.prev_nit, time.perf_counter()))
    sl.prev_nit = eb[:, 35].copy()


t_start = time.perf_counter()
for t in range(T):
    for sl in slabs:
        wait(sl)
        sl.sim.step(sl.act[sl.t], sl.obs, sl.rew, sl.done, sl.info); sl.t += 1; sl.inflight = True
for sl in slabs:
    wait(sl)
wall = time.perf_counter() - t_start
print("G %d x %d envs, %d steps: %.2f ms per batched step (with the read-backs)" % (G, ng, T, wall / T * 1e3))
ev = []
MS = 1e5  # ticks (10 ns) per ms
for sl in slabs:
    for (t, st, en, pn, tw) in sl.rows:
        en = np.where(en < st, en + (1 << 31), en)
        if t < 4:
            continue
        k0, k1 = st.min(), en.max()
        last = int(np.argmax(en))
        dur = en - st
        mw = (pn >= K) if pn is not None else np.zeros(ng, bool)
        lg = int(np.argmax(dur))
        slow = dur > 2.4 * MS
        print("slab %d step %2d: kernel span %.2f ms | last env %4d: started +%.2f, ran %.2f%s | longest env ran %.2f (started +%.2f)%s | envs > 2.4 ms: %d, of them started later than +0.5 ms: %d | 50 / 90 / 100 %% of the envs started by +%.2f / %.2f / %.2f | multi-wave envs %d" % (
            sl.g, t, (k1 - k0) / MS, last, (st[last] - k0) / MS, dur[last] / MS, " MW" if mw[last] else "", dur[lg] / MS, (st[lg] - k0) / MS, " MW" if mw[lg] else "",
            slow.sum(), (slow & (st - k0 > 0.5 * MS)).sum(), np.percentile(st - k0, 50) / MS, np.percentile(st - k0, 90) / MS, (st - k0).max() / MS, mw.sum()))
        ev.append(np.stack([st, np.ones(ng)], 1)); ev.append(np.stack([en, -np.ones(ng)], 1))
ev = np.concatenate(ev); ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1]); dtk = np.diff(ev[:, 0], append=ev[-1, 0])
span = ev[-1, 0] - ev[0, 0]
print("steps >= 4: envs resident (one-wave or multi-wave main) mean %.0f max %d over %.1f ms; time with fewer than 1024 resident: %.0f %%, fewer than 512: %.0f %%, fewer than 128: %.0f %%" % (
    (conc * dtk).sum() / span, conc.max(), span / MS, 100 * dtk[conc < 1024].sum() / span, 100 * dtk[conc < 512].sum() / span, 100 * dtk[conc < 128].sum() / span))

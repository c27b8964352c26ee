"""development (round 6, third session): does an env run slower when its slab shares the chip with three others?
The -DFSIM_TIMELINE build stamps every env's start and end with the device-wide 100 MHz counter.  Slab 0 (envs 0..1023, the same seeds and
actions in both layouts, and an env's bits depend on the env alone) is stepped T times (a) alone on the chip, (b) as one of G slabs stepped
by their own host threads as bench.py does; per step the durations of the SAME envs are compared.
usage: LAYOUT=solo solo_vs_four.py [T]; LAYOUT=shared solo_vs_four.py [T]; solo_vs_four.py [T]   (G, NG, OUT from the environment; FSIM_LIB = the timeline build)"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ.setdefault("FSIM_LIB", os.path.join(ROOT, "furniture_amd", "csrc", "libfsim_tl.so"))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config

m = load_compiled("Sawyer", "table_lack_0825")
G = int(os.environ.get("G", "4")); ng = int(os.environ.get("NG", "1024"))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = default_config(); cfg.max_episode_steps = 150
MS = 1e5  # ticks (10 ns) per ms


def make(g):
    class S: pass
    sl = S(); sl.g = g
    sl.sim = FSim(m, ng, config=cfg)
    sl.sim.set_reset_tables(*ResetTableSampler(m, make_config(), 123, g * ng, ng).draw())
    dev = sl.sim.device
    sl.obs = torch.zeros((ng, sl.sim.obs_dim), device=dev); sl.rew = torch.zeros(ng, device=dev); sl.done = torch.zeros(ng, dtype=torch.uint8, device=dev)
    sl.info = torch.zeros((ng, INFO_DIM), dtype=torch.int32, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(123 + g)
    sl.act = torch.empty((T, ng, 9), device=dev).uniform_(-1, 1, generator=gen)
    sl.sim.reset(None, sl.obs); sl.sim.sync()
    sl.rows = []
    return sl


stop = threading.Event()


def run(sl, record):
    t = -1
    while True:
        t += 1
        if record and t >= T: stop.set(); return
        if not record and stop.is_set(): return   # (the other slabs keep stepping -- their actions over again -- until slab 0 is through)
        h0 = time.perf_counter(); sl.sim.step(sl.act[t % T], sl.obs, sl.rew, sl.done, sl.info); sl.sim.sync(); h1 = time.perf_counter()
        if record:  # (the read-back sits between two steps of this slab only; the other slabs keep running)
            p = sl.sim.get_state("qacc")["qacc"].view(torch.int32)[:, :48].cpu().numpy().astype(np.int64)
            eb = np.ascontiguousarray(sl.sim.get_state("env_block")["env_block"].cpu().numpy()).view(np.int32)
            st, en = p[:, 37], p[:, 38]
            en = np.where(en < st, en + (1 << 31), en)
            sl.rows.append((st.copy(), en.copy(), eb[:, 35].copy(), (h1 - h0) * 1e3))


def layout(n_slabs):
    slabs = [make(g) for g in range(n_slabs)]
    torch.cuda.synchronize(); stop.clear()
    th = [threading.Thread(target=run, args=(sl, sl.g == 0)) for sl in slabs]
    for x in th: x.start()
    for x in th: x.join()
    rows = slabs[0].rows
    for sl in slabs: sl.sim.close() if hasattr(sl.sim, "close") else None
    return rows


# (one layout per process: a handle created and closed before the others shifts HIP's stream -> hardware-queue assignment, and two slabs
#  that share a queue serialise -- seen in the first version of this script: host time per step 8.9 ms = two kernels)
OUT = os.environ.get("OUT", "/tmp/solo_vs_four")
if os.environ.get("LAYOUT") in ("solo", "shared"):
    rows = layout(1 if os.environ["LAYOUT"] == "solo" else G)
    np.savez(OUT + "_" + os.environ["LAYOUT"] + ".npz", st=np.stack([r[0] for r in rows]), en=np.stack([r[1] for r in rows]), nit=np.stack([r[2] for r in rows]), host=np.array([r[3] for r in rows]))
    sys.exit(0)
def load(tag):
    z = np.load(OUT + "_" + tag + ".npz"); return [(z["st"][t], z["en"][t], z["nit"][t], float(z["host"][t])) for t in range(len(z["host"]))]
solo, many = load("solo"), load("shared")
print("slab 0 = envs 0..%d, %d steps; alone on the chip against one of %d slabs (each on its own host thread)" % (ng - 1, T, G))
print("step | Newton iterations equal | kernel span alone / shared (ms) | slowest env alone / shared (ms), ratio | the 16 longest envs: mean ratio | envs of 50 iterations (nothing touches the robot): median ms alone / shared, ratio | first start of the 16 longest, shared (ms after the slab's first env)")
acc = []
for t in range(4, T):
    (s0, e0, n0, h0), (s1, e1, n1, h1) = solo[t], many[t]
    d0, d1 = (e0 - s0) / MS, (e1 - s1) / MS
    top = np.argsort(-d0)[:16]
    quiet = n0 == 50
    r_top = float(np.mean(d1[top] / d0[top]))
    r_q = float(np.median(d1[quiet]) / np.median(d0[quiet])) if quiet.any() else float("nan")
    acc.append(((e0.max() - s0.min()) / MS, (e1.max() - s1.min()) / MS, d0.max(), d1.max(), r_top, r_q, float(np.mean((s1[top] - s1.min()) / MS)), h0, h1))
    if t < 16 or t % 8 == 0:
        print("%3d | %s | %.2f / %.2f | %.2f / %.2f  %.3f | %.3f | %.2f / %.2f  %.3f | %.2f" % (
            t, bool((n0 == n1).all()), acc[-1][0], acc[-1][1], d0.max(), d1.max(), d1.max() / d0.max(), r_top, float(np.median(d0[quiet])), float(np.median(d1[quiet])), r_q, acc[-1][6]))
a = np.array(acc)
print("mean over steps 4..%d: kernel span alone %.3f ms, shared %.3f ms | slowest env alone %.3f, shared %.3f (x %.3f) | 16 longest envs x %.3f | quiet envs x %.3f | the 16 longest start %.3f ms after the slab's first env when shared" % (
    T - 1, a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(), a[:, 3].mean() / a[:, 2].mean(), a[:, 4].mean(), np.nanmean(a[:, 5]), a[:, 6].mean()))
print("host time of fsim_step + fsim_sync (k_schedule + step kernel + launch and wake-up latencies): alone %.3f ms, shared %.3f ms; minus the span of the envs: alone %.3f ms, shared %.3f ms" % (
    a[:, 7].mean(), a[:, 8].mean(), (a[:, 7] - a[:, 0]).mean(), (a[:, 8] - a[:, 1]).mean()))
# who ends the launch when the chip is shared: a four-wave team (the env's previous step took >= K iterations) or a one-wave env, and when it started
K = int(os.environ.get("FSIM_MW_K", "150"))
n_team = n_one = 0; off_last = []; off_team = []; off_one = []; late_top = []
for t in range(5, T):
    s1, e1, n1, _ = many[t]; prev = many[t - 1][2]
    team = prev >= K
    last = int(np.argmax(e1)); k0 = s1.min()
    n_team += bool(team[last]); n_one += not bool(team[last])
    off_last.append((s1[last] - k0) / MS)
    d1 = (e1 - s1) / MS
    top = np.argsort(-d1)[:16]
    off_team += list((s1[top][team[top]] - k0) / MS); off_one += list((s1[top][~team[top]] - k0) / MS)
print("shared layout, steps 5..%d: the env that ends the launch is a team %d times, a one-wave env %d times; it started %.3f ms after the slab's first env on average (max %.2f); of the 16 longest envs of a step the teams start at +%.3f ms, the one-wave envs at +%.3f ms" % (
    T - 1, n_team, n_one, float(np.mean(off_last)), float(np.max(off_last)), float(np.mean(off_team)) if off_team else float("nan"), float(np.mean(off_one)) if off_one else float("nan")))

"""Per-phase shader-clock breakdown of k_env_step using the -DFSIM_PROFILE build (development aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FSIM_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "furniture_amd", "csrc", "libfsim_prof.so"))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config
m = load_compiled("Sawyer", "table_lack_0825")
N = int(os.environ.get("FSIM_PROF_N", "2048"))  # (one slab of the benchmark: handles of up to 2048 envs use the multi-wave scheduling)
cfg = default_config(); cfg.max_episode_steps = 150; cfg.solver_tolerance = float(os.environ.get('FSIM_TOL', '1e-6'))
sim = FSim(m, N, config=cfg)
sampler = ResetTableSampler(m, make_config(), 123, 0, N)
sim.set_reset_tables(*sampler.draw())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev); g = torch.Generator(device=dev); g.manual_seed(123)
sim.reset(None, obs); sim.sync()
names = ["kin+crb+factor", "collide", "vel+smooth", "constraints", "solve"]
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    t0 = time.time(); sim.step(act, obs, rew, done, info); sim.sync(); dt = time.time() - t0
    st_ = sim.get_state("qacc", "contact_geoms")  # (the profile words lie over the debug row: qacc, and behind it contact_geoms)
    pall = torch.cat([st_["qacc"].view(torch.int32), st_["contact_geoms"].reshape(N, -1).view(torch.int32)], 1)[:, :48].cpu().numpy().astype(np.int64)
    p = pall[:, :16]
    cyc = p[:, :5] * 16
    tot = cyc.sum(axis=1)
    nsub, nit, ncoup, nsurv, nslot, maxit = p[:, 5], p[:, 6], p[:, 7], p[:, 8], p[:, 9], p[:, 10]
    print("step %2d: %.1f ms | mean Mcyc/env %.2f (p50 %.2f p99 %.2f max %.2f) | %s | substeps %.0f newton it/substep %.2f (max per substep %d) coupled frac %.3f surv/substep %.1f slots %.1f" % (
        t, dt * 1e3, tot.mean() / 1e6, np.percentile(tot, 50) / 1e6, np.percentile(tot, 99) / 1e6, tot.max() / 1e6,
        " ".join("%s %.0f%%" % (n, 100 * cyc[:, i].sum() / tot.sum()) for i, n in enumerate(names)),
        nsub.mean(), nit.sum() / max(1, nsub.sum()), maxit.max(), ncoup.sum() / max(1, nsub.sum()), nsurv.sum() / max(1, nsub.sum()), nslot.sum() / max(1, nsub.sum())))
    # which envs the scheduler gave four waves in THIS step (its rule, recomputed: last step's Newton iterations >= K; no cap since round 4)
    K = int(os.environ.get("FSIM_MW_K", "150"))
    if t > 0 and os.environ.get("FSIM_MW", "1") not in ("0", "all") and N <= 2048:
        selm = prev_nit >= K
        print("    multi-wave envs %d (%.1f%%): max %.2f mean %.2f Mcyc | one-wave envs: max %.2f Mcyc, %d above 5 Mcyc, %d above 7 Mcyc" % (
            selm.sum(), 100 * selm.mean(), tot[selm].max() / 1e6 if selm.any() else 0, tot[selm].mean() / 1e6 if selm.any() else 0,
            tot[~selm].max() / 1e6, (tot[~selm] > 5e6).sum(), (tot[~selm] > 7e6).sum()))
    eb = sim.get_state("env_block")["env_block"].cpu().numpy()
    prev_nit = np.ascontiguousarray(eb).view(np.int32)[:, 35].copy()
    fine = pall[:, 16:29] * 16
    cfine = pall[:, 29:32] * 16
    fn = ["kinematics", "com_inertia", "crb+M", "factor", "vel_bias", "smooth", "solve:setup", "grad", "hessian", "chol", "Mp+jp", "linesearch", "update+cost"]
    if t in (0, 5):
        med = np.median(fine, axis=0) / 50
        print("    per-substep kcycles (median env): " + " ".join("%s %.1f" % (n, v / 1e3) for n, v in zip(fn, med)) + " | collide %.1f (geom %.1f broad %.1f narrow %.1f) constraints %.1f" % (np.median(cyc[:, 1]) / 50e3, np.median(cfine[:, 0]) / 50e3, np.median(cfine[:, 1]) / 50e3, np.median(cfine[:, 2]) / 50e3, np.median(cyc[:, 3]) / 50e3))
    if t == 5:
        print("    line search: evaluations per Newton iteration: all envs %.2f, slow (top 64) %.2f" % (
            pall[:, 13].sum() / max(1, pall[:, 14].sum()), pall[np.argsort(-tot)[:64], 13].sum() / max(1, pall[np.argsort(-tot)[:64], 14].sum())))
        for e in np.argsort(-tot)[:3]:
            print("    SLOW env %d: Newton solves with a big island %d, mean size %.1f; solves with cached body pairs %d, with uncached (multi-pass) pairs %d" % (e, pall[e, 36] & 0x3ff, pall[e, 35] / max(1, pall[e, 36] & 0x3ff), pall[e, 36] >> 24, (pall[e, 36] >> 16) & 0xff))
            print("    SLOW env %d: Mcyc %.1f it/sub %.2f coupled %.2f | per-substep kcyc: " % (e, tot[e] / 1e6, nit[e] / max(1, nsub[e]), ncoup[e] / max(1, nsub[e])) + " ".join("%s %.1f" % (n, v / 50e3) for n, v in zip(fn, fine[e])) + " | collide %.1f constraints %.1f" % (cyc[e, 1] / 50e3, cyc[e, 3] / 50e3))
    if t == 5:
        for e in list(np.argsort(-tot)[:4]) + [int(np.argsort(tot)[N // 2])]:
            print("    env %d Hessian split kcyc/substep: zero+contact blocks %.1f composite %.1f tree projection %.1f body pairs %.1f limits+welds %.1f" % (
                e, pall[e, 32] * 16 / 50e3, pall[e, 33] * 16 / 50e3, pall[e, 34] * 16 / 50e3, pall[e, 37] * 16 / 50e3, pall[e, 38] * 16 / 50e3))
    if t == 5 and os.environ.get("FSIM_MW") == "all":
        for e in list(np.argsort(-tot)[:3]) + [int(np.argsort(tot)[N // 2])]:
            q = pall[e, 32:42] * 16 / max(1, nit[e]) / 1e3
            print("    env %d multi-wave iteration, main wave kcyc per Newton iteration: stage+post+zero+K %.2f wait[2] %.2f wrench atomics %.2f wait[3] %.2f J'f+norm %.2f wait[4] %.2f project %.2f wait[5] %.2f" % (
                e, q[0], q[1], q[2], q[5], q[6], q[7], q[8], q[9]))
            h = pall[e, 42:48] * 16 / max(1, nit[e]) / 1e3
            print("    env %d helpers between barriers [1] and [3], kcyc per Newton iteration: helper 1 zero %.2f state+blocks %.2f wait %.2f | helper 3 zero %.2f state+blocks %.2f wait %.2f" % (e, h[0], h[1], h[2], h[3], h[4], h[5]))
    if t in (3, 8):
        order = np.argsort(-tot)[:5]
        for e in order:
            print("    slow env %d: Mcyc %.2f phases %s it/sub %.2f coupled %.2f surv %.1f" % (e, tot[e] / 1e6, (cyc[e] / 1e6).round(2), nit[e] / max(1, nsub[e]), ncoup[e] / max(1, nsub[e]), nsurv[e] / max(1, nsub[e])))

"""Which state-based rule picks the envs whose NEXT step is expensive?  Records, per env and step, the robot-part clearance at the
end of the step (E_CLEARANCE) and the step's Newton iterations (E_NITER), then scores rules `clearance < r or niter >= k`:
selected fraction, and the largest iteration count among the envs the rule leaves in the one-wave kernel."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config

E_CLEAR, E_NITER = 34, 35
m = load_compiled("Sawyer", "table_lack_0825")
N, T = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 170
cfg = default_config(); cfg.max_episode_steps = 150
sim = FSim(m, N, config=cfg)
sampler = ResetTableSampler(m, make_config(), 123, 0, N)
sim.set_reset_tables(*sampler.draw())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev); g = torch.Generator(device=dev); g.manual_seed(123)
sim.reset(None, obs); sim.sync()
clear = np.zeros((T, N), np.float32); nit = np.zeros((T, N), np.int64); touch = np.zeros((T, N), np.int64); both = np.zeros((T, N), np.int64); isl = np.zeros((T, N), np.int64)
for t in range(T):
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info); sim.sync()
    if sim.tables_needed() > 0:
        mask = (info[:, 7] > 0).cpu().numpy().astype(np.uint8)
        pq, nz = sampler.draw()
        sim.set_reset_tables(pq, nz, mask=mask)
    eb = sim.get_state("env_block")["env_block"].cpu().numpy()
    eb = np.ascontiguousarray(eb).view(np.int32); clear[t] = eb[:, E_CLEAR].copy().view(np.float32); nit[t] = eb[:, E_NITER]; touch[t] = (eb[:, 18] | eb[:, 19]) != 0; both[t] = (eb[:, 18] & eb[:, 19]) != 0; isl[t] = np.array([bin(int(x) & 0xffff).count('1') for x in eb[:, 20]]) > 1
print("steps %d envs %d | Newton iterations per step: median %.0f p90 %.0f p99 %.0f max %d" % (T, N, np.median(nit), np.percentile(nit, 90), np.percentile(nit, 99), nit.max()))
for t in (0, 5, 20, 60, 100, 140, 149, 150, 160):
    if t < T: print("  step %3d: clearance p10 %.3f p50 %.3f | <2cm %.3f <5cm %.3f <10cm %.3f | niter>=100: %.3f >=150: %.3f >=200: %.3f" % (
        t, np.percentile(clear[t], 10), np.median(clear[t]), (clear[t] < .02).mean(), (clear[t] < .05).mean(), (clear[t] < .10).mean(), (nit[t] >= 100).mean(), (nit[t] >= 150).mean(), (nit[t] >= 200).mean()))
nxt = nit[1:]; cl = clear[:-1]; prev = nit[:-1]; tc = touch[:-1] > 0; bt = both[:-1] > 0; cp = isl[:-1] > 0
print("touch-any frac %.3f, touch-both frac %.3f, robot-coupled frac %.3f, clearance<1cm %.3f <2cm %.3f <3cm %.3f" % (tc.mean(), bt.mean(), cp.mean(), (cl < .01).mean(), (cl < .02).mean(), (cl < .03).mean()))
def score(name, sel):
    frac = sel.mean(axis=1); rest = np.where(sel, 0, nxt); big = nxt >= 150
    print("  %-44s selected mean %.3f max %.3f | non-selected next iters: max %4d p99.9 %4.0f mean-of-step-max %5.1f | recall(next>=150) %.3f" % (
        name, frac.mean(), frac.max(), rest.max(), np.percentile(rest, 99.9), rest.max(axis=1).mean(), (sel & big).sum() / max(1, big.sum())))
for k in (100, 150, 200):
    score("prev>=%d" % k, prev >= k)
    score("prev>=%d or touch" % k, (prev >= k) | tc)
    score("prev>=%d or coupled" % k, (prev >= k) | cp)
    score("prev>=%d or touch or coupled" % k, (prev >= k) | tc | cp)
    for r in (0.005, 0.01, 0.02, 0.03):
        score("prev>=%d or touch or coupled or clear<%.3f" % (k, r), (prev >= k) | tc | cp | (cl < r))

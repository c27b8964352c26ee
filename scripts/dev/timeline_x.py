"""development: who finishes last in a k_env_step_x launch (-DFSIM_PROFILE -DFSIM_TIMELINE build, FSIM_LIB=...): start / end tick per env,
whether it was stepped by four waves, its Newton iterations.  usage: timeline_x.py [steps] [multi_wave mode]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM, MULTI_WAVE, E_MW_STEPS, E_NITER
E_CLEARANCE, E_TOUCH_L, E_TOUCH_R, E_TOUCH_FLOOR = E_NITER - 1, 18, 19, 20
from furniture_amd.envs import ResetTableSampler, make_config
m = load_compiled("Sawyer", "table_lack_0825")
N = int(os.environ.get("FSIM_PROF_N", "4096"))
cfg = default_config(); cfg.max_episode_steps = 1000; cfg.lookahead_reset = 0
cfg.multi_wave = MULTI_WAVE[sys.argv[2] if len(sys.argv) > 2 else "auto"]
sim = FSim(m, N, config=cfg)
print("kernel", sim.step_kernel)
sim.set_reset_tables(*ResetTableSampler(m, make_config(), 123, 0, N).draw())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev); g = torch.Generator(device=dev); g.manual_seed(123)
sim.reset(None, obs); sim.sync()
mw_prev = np.zeros(N, dtype=np.int64)
dump = []
feat = []
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    t0 = time.time(); sim.step(act, obs, rew, done, info); sim.sync(); dt = time.time() - t0
    p = sim.get_state("qacc")["qacc"].view(torch.int32).cpu().numpy().astype(np.int64)
    eb = sim.get_state("env_block")["env_block"].view(torch.int32).cpu().numpy().astype(np.int64)
    mw = eb[:, E_MW_STEPS] - mw_prev; mw_prev = eb[:, E_MW_STEPS].copy()
    nit = eb[:, E_NITER]
    st, en = p[:, 37].copy(), p[:, 38].copy()
    en = np.where(en < st, en + (1 << 31), en)
    t_0 = st.min(); st -= t_0; en -= t_0
    dump.append(np.stack([st, en, mw, nit]))
    qv = np.concatenate([sim.get_state("qvel")["qvel"].cpu().numpy(), sim.get_state("qpos")["qpos"].cpu().numpy()[:, :9]], axis=1)
    feat.append(np.concatenate([eb[:, [E_TOUCH_L, E_TOUCH_R, E_TOUCH_FLOOR]].astype(np.float32), sim.get_state("env_block")["env_block"][:, E_CLEARANCE:E_CLEARANCE + 1].cpu().numpy(), qv], axis=1).astype(np.float32))
    us = lambda x: x / 100.0  # 100 MHz ticks -> us
    one, four = mw == 0, mw > 0
    d = en - st
    line = "step %2d: wall %.2f ms span %.0f us | one-wave envs %d: duration mean %.0f p99 %.0f max %.0f us, last end %.0f us" % (t, dt * 1e3, us(en.max()), one.sum(), us(d[one].mean()), us(np.percentile(d[one], 99)), us(d[one].max()), us(en[one].max()))
    if four.any():
        line += " | four-wave envs %d: duration mean %.0f max %.0f us, last start %.0f last end %.0f us; first one-wave start %.0f us" % (four.sum(), us(d[four].mean()), us(d[four].max()), us(st[four].max()), us(en[four].max()), us(st[one].min()))
    lc = p[:, 36]
    line += " | record load %.1f us, model-cache build %.1f us (built by %d of the waves)" % (p[:, 35].mean() / 100.0, lc[lc >= 0].mean() / 100.0 if (lc >= 0).any() else 0.0, int((lc >= 0).sum()))
    print(line)
    last = np.argsort(-en)[:8]
    print("     last to finish: " + "  ".join("[env %d %s start %.0f dur %.0f it %d]" % (e, "4w" if four[e] else "1w", us(st[e]), us(d[e]), nit[e]) for e in last))
    # iterations of this step vs the duration, one-wave envs: us per iteration at the top
    top = np.argsort(-d)[:5]
    print("     longest: " + "  ".join("[env %d %s start %.0f dur %.0f it %d]" % (e, "4w" if four[e] else "1w", us(st[e]), us(d[e]), nit[e]) for e in top))
if os.environ.get("FSIM_TL_DUMP"):
    np.save(os.environ["FSIM_TL_DUMP"], np.stack(dump).astype(np.int32))
    np.save(os.environ["FSIM_TL_DUMP"].replace(".npy", "_feat.npy"), np.stack(feat).astype(np.float16))  # [step, env, (touch l, r, floor-island word, clearance, qvel)] after the step  # [step, (start, end, four-wave, iterations), env]

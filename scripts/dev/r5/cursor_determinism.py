"""development: is the device's Cursor path a function of its inputs?  The same reset tables and actions, R runs in one process: per step
the envs whose observation / info rows differ from run 0 bitwise.  usage: cursor_determinism.py <agent> <furniture> <n> <steps> <runs>"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from furniture_amd.envs import ResetTableSampler, make_config
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, INFO_DIM, default_config

agent, furn, n, steps, runs = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
m = load_compiled(agent, furn)
ecfg = make_config(unity=False, record_vid=False, furniture_name=furn, max_episode_steps=1000, seed=200)
parts, noise = ResetTableSampler(m, ecfg, 200, 0, n).draw()
ref = None
for r in range(runs):
    cfg = default_config()
    cfg.max_episode_steps, cfg.auto_reset = 1000, 0
    sim = FSim(m, n, config=cfg)
    sim.set_reset_tables(parts, noise if agent != "Cursor" else None)
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs)
    sim.sync()
    dof = sim.dof_action
    act, rew = torch.zeros((n, dof), device=dev), torch.zeros(n, device=dev)
    done, info = torch.zeros(n, dtype=torch.uint8, device=dev), torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    rng = np.random.RandomState(1)
    trace = [(obs.cpu().numpy().copy(), None)]
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, dof)).astype(np.float32)
        if agent == "Cursor":
            a[:, 6] = np.abs(a[:, 6]) * np.where(rng.rand(n) < 0.8, 1, -1)
            a[:, 13] = np.abs(a[:, 13]) * np.where(rng.rand(n) < 0.8, 1, -1)
        act.copy_(torch.as_tensor(a))
        torch.cuda.synchronize()
        if os.environ.get("POISON"):
            import ctypes
            PL = ctypes.CDLL(os.path.join(ROOT, "tests", "liblds_poison.so"))
            for _ in range(3):
                assert PL.lds_poison(ctypes.c_uint(int(os.environ["POISON"], 16) if r % 2 == 0 else 0)) == 0
        sim.step(act, obs, rew, done, info)
        sim.sync()
        trace.append((obs.cpu().numpy().copy(), info.cpu().numpy()[:, [0, 1, 2, 3, 4, 5, 6, 7, 12, 15, 16]].copy()))
    sim.close()
    if ref is None:
        ref = trace
        print("run 0: kernel %s, fails per step %s, overflow words per step %s" % (sim.step_kernel, [int(x[1][:, 2].sum()) for x in trace[1:]], [int((x[1][:, 8] != 0).sum()) for x in trace[1:]]))
    else:
        diff = []
        for t, ((o0, i0), (o1, i1)) in enumerate(zip(ref, trace)):
            bad = np.nonzero((o0.view(np.uint32) != o1.view(np.uint32)).any(axis=1) | ((i0 != i1).any(axis=1) if i0 is not None else False))[0]
            if len(bad):
                diff.append((t - 1, bad.tolist()[:6]))
        print("run %d: %s" % (r, "identical to run 0" if not diff else "DIFFERS (step, envs): %s" % diff[:6]))

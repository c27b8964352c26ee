"""Does the fp32 device solver spend more Newton iterations than the fp64 oracle on the SAME step?  64 envs in lockstep with 64
oracle envs (same seeds, same actions, as tests/test_contact_stress_gpu.py); per env and step: the device's iteration sum over the
step's 50 substeps (E_NITER) beside the oracle's, for the oracle at tolerance 1e-8 (the reference's) -- while the two still agree."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, INFO_DIM, default_config
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig

E_NITER = 35
m = load_compiled("Sawyer", "table_lack_0825")
n, T = int(os.environ.get("N", 64)), int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = default_config(); cfg.max_episode_steps, cfg.auto_reset = 150, 0
sim = FSim(m, n, config=cfg)
envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=1000 + i, solver_tolerance=1e-8)) for i in range(n)]
acc = [0] * n
for i, e in enumerate(envs):
    def wrap(orig, i, s):
        def step():
            orig()
            acc[i] += s.last_solver_iters
        return step
    e.sim.step = wrap(e.sim.step, i, e.sim)
obs_o = [e.flat_obs(e.reset()) for e in envs]
sim.set_reset_tables(np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]), np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs]))
dev = sim.device
obs = torch.zeros((n, sim.obs_dim), device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev)
info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev); act = torch.zeros((n, sim.dof_action), device=dev)
sim.reset(None, obs); sim.sync()
rng = np.random.RandomState(7)
agree = np.ones(n, bool)
rows = []
for t in range(T):
    a = rng.uniform(-1, 1, (n, sim.dof_action)).astype(np.float32)
    act.copy_(torch.as_tensor(a)); torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info); sim.sync()
    eb = np.ascontiguousarray(sim.get_state("env_block")["env_block"].cpu().numpy()).view(np.int32)
    od = obs.cpu().numpy()
    for i, e in enumerate(envs):
        acc[i] = 0
        ob, r, d, _ = e.step(a[i])
        if agree[i] and np.abs(od[i] - e.flat_obs(ob)).max() > 1e-3: agree[i] = False
        if agree[i]: rows.append((t, i, int(eb[i, E_NITER]), acc[i]))
R = np.array(rows)
print("env-steps compared (still in agreement): %d | device iterations mean %.1f, oracle %.1f | ratio of sums %.3f" % (len(R), R[:, 2].mean(), R[:, 3].mean(), R[:, 2].sum() / R[:, 3].sum()))
for lo, hi in ((0, 60), (60, 100), (100, 150), (150, 250), (250, 10000)):
    s = (R[:, 3] >= lo) & (R[:, 3] < hi)
    if s.any(): print("  oracle iterations in [%d, %d): %5d env-steps, device mean %.1f oracle mean %.1f (device/oracle %.3f), device max %d oracle max %d" % (lo, hi, s.sum(), R[s, 2].mean(), R[s, 3].mean(), R[s, 2].sum() / R[s, 3].sum(), R[s, 2].max(), R[s, 3].max()))
top = R[np.argsort(-R[:, 2])[:12]]
print("  slowest device env-steps (step, env, device, oracle):", [tuple(int(x) for x in r) for r in top])

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FSIM_VERBOSE"] = "1"
import numpy as np, torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, INFO_DIM, default_config
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
m = load_compiled("Sawyer", "table_lack_0825")
n = 2
cfg = default_config(); cfg.max_episode_steps = 150; cfg.auto_reset = 0; cfg.control_type = 7
sim = FSim(m, n, config=cfg)
envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + i, solver_tolerance=1e-10, control_type="ik")) for i in range(n)]
obs_o = [e.reset() for e in envs]
sim.set_reset_tables(np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]), np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs]))
dev = sim.device
obs = torch.zeros((n, sim.obs_dim), device=dev)
sim.reset(None, obs); sim.sync()
print("dof", sim.dof_action, "obs_dim", sim.obs_dim, "reset err", max(np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(obs_o[e])).max() for e in range(n)))
blk = sim.get_state("env_block")["env_block"][:, -26:].cpu().numpy().view(np.float32)
for e in range(n):
    print(" target dev", blk[e, :3], "oracle", envs[e]._ik_target_pos, "iquat dev", blk[e, 3:7], "oracle", envs[e]._initial_right_hand_quat)
dof = sim.dof_action
act = torch.zeros((n, dof), device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev); info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
rng = np.random.RandomState(5)
for t in range(5):
    a = rng.uniform(-1, 1, (n, dof)).astype(np.float32)
    if t < 2: a[:, 3:6] = 0
    act.copy_(torch.as_tensor(a)); torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info); sim.sync()
    blk = sim.get_state("env_block")["env_block"][:, -26:].cpu().numpy().view(np.float32)
    for e in range(n):
        ob, r, d_, _ = envs[e].step(a[e].astype(np.float64))
        d = np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(ob))
        print("  step", t, "env", e, "obs err %.2e at %d" % (d.max(), d.argmax()), "qcmd err %.2e" % np.abs(blk[e, 7:14] - envs[e]._ik_q_cmd).max(),
              "target err %.2e" % np.abs(blk[e, :3] - envs[e]._ik_target_pos).max(), "iquat err %.2e" % np.abs(blk[e, 3:7] - envs[e]._initial_right_hand_quat).max(), "rew", float(rew[e]), r)
sim.close()

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, INFO_DIM, default_config
from furniture_amd.envs import CONTROLLER_CODES
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
from oracle import controllers as C
kinds = sys.argv[1:] or list(CONTROLLER_CODES)
for kind in kinds:
  for zero_applied in (0, 1):
    m = load_compiled("Sawyer", "table_lack_0825", kind)
    n = 2
    cfg = default_config(); cfg.max_episode_steps = 150; cfg.auto_reset = 0; cfg.control_type = CONTROLLER_CODES[kind]
    sim = FSim(m, n, config=cfg)
    envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + i, solver_tolerance=1e-10, control_type=kind)) for i in range(n)]
    obs_o = [e.reset() for e in envs]
    sim.set_reset_tables(np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]), np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs]))
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs); sim.sync()
    print(kind, "zero_applied", zero_applied, "dof", sim.dof_action, "obs_dim", sim.obs_dim, "reset err", max(np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(obs_o[e])).max() for e in range(n)))
    # clean start on both sides: the oracle's poses, arm at rest, applied forces = 0 (zero_applied) or the bias of that pose
    for e in envs:
        e.sim.data.qvel[:] = 0; e.sim.data.qacc_warmstart[:] = 0
        e.sim.data.qpos[m.arm_qposadr] = m.arm_initqpos
        e.sim.forward()
        e.sim.data.qfrc_applied[:] = 0
        if not zero_applied: e._gravity_comp()
    sim.set_state(qpos=np.stack([e.sim.data.qpos for e in envs]), qvel=np.zeros((n, m.nv)), qacc_warmstart=np.zeros((n, m.nv)),
                  qfrc_applied=np.stack([e.sim.data.qfrc_applied for e in envs]))
    dof = sim.dof_action
    act = torch.zeros((n, dof), device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.uint8, device=dev); info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    rng = np.random.RandomState(5)
    for t in range(4):
        a = rng.uniform(-1, 1, (n, dof)).astype(np.float32)
        act.copy_(torch.as_tensor(a)); torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info); sim.sync()
        st = sim.get_state("ctrl", "qpos", "qvel")
        for e in range(n):
            ob, r, d_, _ = envs[e].step(a[e].astype(np.float64))
            d = np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(ob))
            dc = np.abs(st["ctrl"][e].cpu().numpy() - envs[e].sim.data.ctrl)
            print("  step", t, "env", e, "obs err %.2e at %d" % (d.max(), d.argmax()), "ctrl err %.2e (|ctrl| %.1f)" % (dc.max(), np.abs(envs[e].sim.data.ctrl).max()),
                  "qvel max %.2f" % np.abs(envs[e].sim.data.qvel[:7]).max(), "rew", float(rew[e]), r)
    sim.close()

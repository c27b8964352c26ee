import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, INFO_DIM, default_config
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
m = load_compiled("Sawyer", "chair_agne_0007")
cfg = default_config(); cfg.max_episode_steps = 150; cfg.auto_reset = 0
sim = FSim(m, 1, config=cfg)
e = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123, solver_tolerance=1e-10))
ob = e.reset()
sim.set_reset_tables(e.reset_draws["part_qpos"].reshape(1, -1), np.stack(e.reset_draws["noise"]).reshape(1, -1))
obs = torch.zeros((1, sim.obs_dim), device=sim.device)
sim.reset(None, obs); sim.sync()
st = sim.get_state("ncon", "contact_geoms", "qpos", "qvel")
nc = int(st["ncon"][0, 0]); cg = st["contact_geoms"][0].cpu().numpy().reshape(-1, 2)[:nc]
print("device ncon", nc, cg.tolist())
print("oracle ncon", e.sim.ncon, e.sim.contacts())
print("geom types", {int(g): (int(m.geom_type[g]), m.geom_size[g].tolist(), int(m.geom_bodyid[g])) for g in set(cg.reshape(-1).tolist())})
a = m.part_qposadr[2]
print("dev qpos part2", st["qpos"][0, a:a+7].cpu().numpy(), "oracle", e.sim.data.qpos[a:a+7])
d = m.part_dofadr[2]
print("dev qvel part2", st["qvel"][0, d:d+6].cpu().numpy())

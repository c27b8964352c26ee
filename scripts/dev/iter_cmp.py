import sys, numpy as np, torch
sys.path.insert(0,'/root/repo' if __import__('os').path.exists('/root/repo/oracle') else '.')
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
m = load_compiled("Sawyer","table_lack_0825")
sim = FSim(m, 1)
for seed in (124, 137, 138, 142):
    env = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=seed, solver_tolerance=1e-6))
    env.reset(); rng = np.random.RandomState(seed); o = env.sim
    rob = m.geom_is_robot.astype(bool); pc = m.geom_is_partcol.astype(bool)
    for t in range(25):
        env.step(rng.uniform(-1,1,9))
        if any((rob[a] and pc[b]) or (rob[b] and pc[a]) for a,b in o.contacts()): break
    o.set_solver(100, 1e-6, "newton")
    it_o, it_d, dq = [], [], []
    for s in range(150):
        # copy oracle state to device
        sim.set_state(qpos=o.data.qpos[None].copy(), qvel=o.data.qvel[None].copy(), qacc_warmstart=o.data.qacc_warmstart[None].copy(),
                      ctrl=o.data.ctrl[None].copy(), qfrc_applied=o.data.qfrc_applied[None].copy(),
                      xfrc_applied=o.data.xfrc_applied[[int(b) for b in m.part_bodyid]].reshape(1,-1).copy(),
                      geom_contype=o.model.geom_contype[None].copy(), geom_conaffinity=o.model.geom_conaffinity[None].copy(),
                      eq_active=o.model.eq_active[None].copy(), eq_data=o.model.eq_data.reshape(1,-1).copy())
        sim.physics_step(1)
        st = sim.get_state("solver_iters", "qacc")
        o.step()
        it_o.append(o.last_solver_iters); it_d.append(int(st["solver_iters"][0,0]))
        dq.append(np.abs(st["qacc"][0].cpu().numpy() - o.data.qacc).max() / (1 + np.abs(o.data.qacc).max()))
    print("seed", seed, "oracle it/substep %.2f (max %d)  device %.2f (max %d)  qacc rel err median %.1e max %.1e" % (np.mean(it_o), max(it_o), np.mean(it_d), max(it_d), np.median(dq), max(dq)))

import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import tests.test_demo_sawyer_replay as T
from furniture_amd.sim import FSim, default_config
for name in sorted(T.SEGMENTS):
    m = T.kinematic_robot_model(); sim = FSim(m, 2, config=default_config())
    f0, f1 = T.SEGMENTS[name]; q = T.start_state(m, f0); zero = lambda k: np.zeros((1,k))
    sim.set_state(qpos=q[None], qvel=zero(m.nv), qacc_warmstart=zero(m.nv), ctrl=zero(m.nu), qfrc_applied=zero(m.nv), xfrc_applied=zero(6*m.nparts))
    traj=[]
    for t in range(f0, f1):
        v = (T.robot(t+1)-T.robot(t))/(T.N_SUB*T.H)
        st = sim.get_state("qpos","qvel"); qp, qv = st["qpos"].clone(), st["qvel"].clone()
        qp[:, :9] = torch.as_tensor(T.robot(t), dtype=torch.float32, device=qp.device); qv[:, :9] = torch.as_tensor(v, dtype=torch.float32, device=qp.device)
        sim.set_state(qpos=qp, qvel=qv, ctrl=np.concatenate([v[:7], T.robot(t+1)[7:9]])[None])
        sim.physics_step(T.N_SUB); sim.sync()
        qn = sim.get_state("qpos")["qpos"][0].cpu().numpy().astype(np.float64)
        traj.append(np.array([qn[int(a):int(a)+7] for a in m.part_qposadr]))
    sim.close()
    ora = T.replay_oracle(name)
    E=[T.errors(p, f0+k+1) for k,p in enumerate(traj)]; dp=np.array([e[0] for e in E]); dq=np.array([e[1] for e in E])
    d = np.array([np.abs(a[:, :3]-b[:, :3]).max() for a,b in zip(traj, ora)])
    print(name, "dp max per part", dp.max(0).round(4), "dq max", dq.max(0).round(3))
    print("  vs oracle d[:20]", d[:20].round(4), "max", d.max().round(4), "argmax", d.argmax())
    if name.startswith("hold"): print("  col dp[:8]", dp[:8,1].round(4), "dq", dq[:8,1].round(3), "z", [round(p[1,2],4) for p in traj[3:8]])
    else: print("  seat dp[:30]", dp[:30,2].round(4), "25:60 max", dp[25:60,2].max())

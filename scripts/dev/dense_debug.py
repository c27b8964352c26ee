import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from furniture_amd.dense import pack_dense

def main():
    import torch
    from furniture_amd.mjcf.model import load_compiled
    from furniture_amd.sim import FSim, INFO_DENSE_PHASE, INFO_DIM, default_config
    from oracle.dense_reward import DenseConfig
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    from tests.scenarios import counter_actions, pinch_attach_state

    m = load_compiled("Sawyer", "table_lack_0825")
    n = 2
    # the base env's lenient alignment thresholds and a lenient eef_rot_threshold, so that the scripted pinch below (arm pose
    # as left by the reset, gripper ~30 deg off vertical) walks the state machine: early pick -> lift_leg, connect -> next subtask
    kw = dict(max_episode_steps=150, auto_align=False)
    cfg = default_config()
    cfg.max_episode_steps, cfg.auto_reset, cfg.auto_align, cfg.dense_reward = 150, 0, 0, 1
    sim = FSim(m, n, config=cfg)
    sim.set_dense_reward(*pack_dense(m, dict(eef_rot_threshold=0.8)))
    envs = [FurnitureEnvOracle(m, OracleConfig(seed=321 + i, solver_tolerance=1e-10, dense=DenseConfig(eef_rot_threshold=0.8), **kw))
            for i in range(n)]
    obs_o = [e.reset() for e in envs]
    sim.set_reset_tables(np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]),
                         np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs]))
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs)
    sim.sync()
    act = torch.zeros((n, 9), device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev)
    info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)

    def resync():
        st = sim.get_state("qpos", "qvel", "qacc_warmstart", "dense")
        for e in range(n):
            d = envs[e].sim.data
            d.qpos[:] = st["qpos"][e].cpu().numpy(); d.qvel[:] = st["qvel"][e].cpu().numpy(); d.qacc_warmstart[:] = st["qacc_warmstart"][e].cpu().numpy()
            ds_ = st["dense"][e].cpu().numpy().astype(np.float64)
            D = envs[e]._dense
            D.init_table_site_pos, D.init_lift_leg_pos, D.lift_leg_pos, D.init_eef_pos = ds_[4:7].copy(), ds_[7:10].copy(), ds_[10:13].copy(), ds_[13:16].copy()
            (D.prev_init_eef_dist, D.prev_eef_above_leg_dist, D.prev_eef_leg_dist, D.prev_grasp_dist, D.prev_lift_leg_z_dist, D.prev_lift_leg_xy_dist,
             D.prev_move_pos_dist, D.prev_move_up_ang_dist, D.prev_move_forward_ang_dist, D.prev_proj_t, D.prev_proj_l) = ds_[16:27]

    def both(a, sync=False):
        if sync:
            resync()
        act.copy_(torch.as_tensor(np.asarray(a, dtype=np.float32)))
        torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info)
        sim.sync()
        out = []
        for e in range(n):
            ob, r, d, inf = envs[e].step(np.asarray(a[e], dtype=np.float64))
            # reward terms scale distances by up to 1e4: 2e-6 m of fp32 physics noise -> 2e-2
            od = np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(ob))
            print("env", e, "rew dev %.4f oracle %.4f diff %.4f | done %d/%d phase %d/%d | obs maxdiff %.2e at %d | dense dev" % (float(rew[e]), r, float(rew[e]) - r, int(done[e]), d, int(info[e, INFO_DENSE_PHASE]), inf["phase_i"], od.max(), od.argmax()), sim.get_state("dense")["dense"][e].cpu().numpy().round(4)[[1,2,20,21]], "oracle prev z/xy", envs[e]._dense.prev_lift_leg_z_dist, envs[e]._dense.prev_lift_leg_xy_dist)
            out.append((r, d, inf))
        return out

    for t in range(6):
        both(np.stack([counter_actions(321, i, t, 9) for i in range(n)]))
    # scripted: the gripper pinches the leg of recipe step 0 (part 1, connector 1) next to its table connector (5)
    for e in range(n):
        o = envs[e]
        q, xfrc, masks = pinch_attach_state(m, o.sim.data.qpos.copy(), o.sim.data.xpos.copy(), o.sim.data.xquat.copy(), leg=1,
                                            table_conn=5, leg_conn=1, gap=0.02)
        o.sim.data.qpos[:], o.sim.data.qvel[:], o.sim.data.qacc_warmstart[:] = q, 0, 0
        for i in range(m.nparts):
            o.sim.data.xfrc_applied[m.part_bodyid[i]] = xfrc.reshape(-1, 6)[i]
        for g, (ct, ca) in masks.items():
            o.sim.model.geom_contype[g], o.sim.model.geom_conaffinity[g] = ct, ca
        if e == 0:
            Q, X, M = [q], [xfrc], masks
        else:
            Q.append(q); X.append(xfrc)
    gm = sim.get_state("geom_contype", "geom_conaffinity")
    for g, (ct, ca) in M.items():
        gm["geom_contype"][:, g], gm["geom_conaffinity"][:, g] = ct, ca
    sim.set_state(qpos=np.stack(Q), qvel=np.zeros((n, m.nv)), qacc_warmstart=np.zeros((n, m.nv)), xfrc_applied=np.stack(X),
                  geom_contype=gm["geom_contype"], geom_conaffinity=gm["geom_conaffinity"])
    # the scenario teleported the table: re-anchor the reward's "table must not move" reference on both sides (the device
    # state block is part of the snapshot: state field 'dense', ED_INIT_TABLE_SITE = 4..6)
    ds = sim.get_state("dense")["dense"]
    for e in range(n):
        envs[e].sim.forward()
        tsite = envs[e].sim.data.site_xpos[envs[e]._dsub[0]["table_site"]].copy()
        envs[e]._dense.init_table_site_pos = tsite
        ds[e, 4:7] = torch.as_tensor(tsite, dtype=torch.float32)
    sim.set_state(dense=ds)
    a = np.zeros((n, 9), dtype=np.float32)
    a[:, 7] = 1.0   # close the gripper, do not connect yet
    a[:, 8] = -1.0
    seen = set()
    for t in range(3):
        for (_, _, inf) in both(a, True):
            seen.add(inf["phase_i"] % 8)
    a[:, 8] = 1.0   # connect
    res = both(a, True)
    for (_, _, inf) in res:
        seen.add(inf["phase_i"] % 8)
    print("phases visited:", sorted(seen), "after connect:", [(r[2]["phase_i"], r[2]["num_connected"], r[2]["subtask"]) for r in res])
    print(all(r[2]["num_connected"] == 1 and r[2]["subtask"] == 1 for r in res) and 4 in seen)
    for t in range(2):
        both(a, True)
    sim.close()


main()

"""oracle/dense_reward.py against the reference's own FurnitureSawyerDenseRewardEnv._compute_reward (golden vectors from
scripts/make_golden_dense.py: 48 scripted episodes through all 8 phases and 4 subtasks, incl. drops, table moves, wrong
connects, early termination, preassembled starts)."""
import os

import numpy as np

from oracle.dense_reward import DenseConfig, DenseReward

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dense_reward.npz"))


def subtasks_from_golden():
    subs = []
    for i in range(len(G["angles"])):
        gi = G["grip_init"][i]
        gi = None if np.isnan(gi[0]) else [float(x) for x in gi if not np.isnan(x)]
        ang = G["angles"][i]
        subs.append(dict(angle=None if np.isnan(ang) else float(ang), has_angles=bool(G["has_angles"][i]),
                         waypoint_z=float(G["waypoint_z"][i]), grip_init=gi))
    return subs


def test_dense_reward_matches_reference():
    subs = subtasks_from_golden()
    nsteps = 0
    phases = set()
    for e in range(int(G["n_ep"])):
        diff, early, rra, n_pre = G["ep%d_flags" % e]
        cfg = DenseConfig(diff_rew=bool(diff), early_termination=bool(early), reset_robot_after_attach=bool(rra))
        dr = DenseReward(cfg, subs, z_finedist=0.05, success_num_conn=len(subs), n_pre=int(n_pre))
        cur = {"o": G["ep%d_obs0" % e]}
        dr.reset(lambda st: cur["o"][st])
        obs, ac, conn = G["ep%d_obs" % e], G["ep%d_ac" % e], G["ep%d_connected" % e]
        for t in range(len(ac)):
            cur["o"] = obs[t]
            r, done, succ, info = dr.compute(ac[t], lambda st: bool(cur["o"][st][38]), bool(conn[t]))
            ref_r = G["ep%d_reward" % e][t]
            assert abs(r - ref_r) <= 1e-9 * max(1.0, abs(ref_r)), (e, t, r, ref_r)
            assert done == G["ep%d_done" % e][t] and succ == G["ep%d_success" % e][t], (e, t)
            assert dr.phase_i == G["ep%d_phase" % e][t] and dr.subtask_step == G["ep%d_subtask" % e][t], (e, t)
            assert abs(info["phase_bonus"] - G["ep%d_phase_bonus" % e][t]) < 1e-9
            phases.add(int(dr.phase_i))
            nsteps += 1
    assert nsteps > 3000 and phases == set(range(8))

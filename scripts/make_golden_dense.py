"""Generate tests/golden/dense_reward.npz by driving the REFERENCE's FurnitureSawyerDenseRewardEnv._compute_reward /
_reset_reward_variables (furniture_sawyer_dense.py:128-577) on an instance created without __init__, whose sensor getters
(_get_pos, _get_up_vector, _get_forward_vector, _finger_contact, _is_aligned) read a scripted kinematic "puppet" scene.

Runs only in the build container (needs /root/reference); see scripts/make_golden_env_logic.py for the stub-module import."""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from make_golden_env_logic import axis_rot, import_reference, rand_rot  # noqa: E402

from oracle.dense_reward import O_DIM, DenseConfig  # noqa: E402  (only the config defaults + the layout constants)

N_SUB = 4
ANGLES = [90.0, 90.0, 270.0, None]


def main():
    import_reference()
    import furniture.env.furniture_sawyer_dense as D
    D.logger = types.SimpleNamespace(info=lambda *a, **k: None, warn=lambda *a, **k: None, debug=lambda *a, **k: None)
    Ref = D.FurnitureSawyerDenseRewardEnv

    class Scene:
        pass

    class Fake(Ref):
        def __init__(self):  # no sim
            self._unity = None

        def _get_pos(self, name):
            return np.array(self.scene.pos[name], dtype=float)

        def _get_up_vector(self, name):
            return self.scene.rot[name][:, 2].copy()

        def _get_forward_vector(self, name):
            return self.scene.rot[name][:, 1].copy()

        def _finger_contact(self, leg):
            return self.scene.touch[leg]

        def _is_aligned(self, a, b):
            return self.scene.aligned[a]

    rng = np.random.RandomState(5)
    episodes = []
    for ep in range(48):
        cfg = DenseConfig()
        cfg.diff_rew = True  # the reference itself raises AttributeError in grasp_leg when diff_rew is False (:668)
        cfg.early_termination = ep % 5 == 4
        cfg.reset_robot_after_attach = ep % 7 == 6
        n_pre = 1 if ep % 6 == 5 else 0
        f = Fake()
        f._config = types.SimpleNamespace(**{k: v for k, v in vars(cfg).items()})
        for k, v in vars(cfg).items():
            setattr(f, "_" + k, v)
        legs = ["leg%d" % i for i in range(N_SUB)]
        leg_sites = ["leg%d-table,0,90,180,270,conn_site" % i if ANGLES[i] is not None or i != 3 else "leg3-table,conn_site" for i in range(N_SUB)]
        leg_sites[3] = "leg3-table,conn_site"  # one subtask without allowed angles: forward is not projected
        table_sites = ["table-leg%d,0,90,180,270,conn_site%d" % (i, i) for i in range(N_SUB)]
        # subtask 2 exercises the angle=None branch of _project_connector_forward with allowed angles present
        site_recipe = [[leg_sites[i], table_sites[i]] + ([ANGLES[i]] if i in (0, 1) else []) for i in range(N_SUB)]
        grip_init = [None, [[0, 0, 0, 0.37]], [[0.02, -0.01, 0.03]], [[0, 0, 0, 0.37]]]
        wz = [0.1, 0.1, 0.12, 0.08]
        f._recipe = dict(recipe=[[legs[i], "table"] for i in range(N_SUB)], waypoints=[[[0, 0, wz[i]]] for i in range(N_SUB)],
                         grip_init_pos=grip_init, z_finedist=0.05)
        f._site_recipe = site_recipe
        f._object_name2id = {n: i for i, n in enumerate(legs + ["table"])}
        f._preassembled = list(range(n_pre))
        f._success_num_conn = N_SUB
        f._phases = ["init_eef", "move_eef_above_leg", "lower_eef", "grasp_leg", "lift_leg", "align_leg", "move_leg", "move_leg_fine"]
        f._grip_up_phases = set(f._phases[:5])
        f._grip_forward_phases = set(f._phases[1:5])
        f._grip_open_phases = set(f._phases[:3])
        f._max_episode_steps = 10 ** 9
        f._episode_length = 0
        f._connected = False
        f._success = False
        sc = f.scene = Scene()
        # ---- scene
        tableR = axis_rot(np.array([0, 0, 1.0]), rng.uniform(-0.3, 0.3))
        table_pos = np.array([0.0, 0.0, 0.02])
        site_off = [np.array([sx * 0.2, sy * 0.15, 0.02]) for sx in (-1, 1) for sy in (-1, 1)]
        legR = [axis_rot(np.array([0, 0, 1.0]), rng.uniform(-3, 3)) @ axis_rot(np.array([1.0, 0, 0]), np.pi / 2) for _ in range(N_SUB)]
        leg_pos = [np.array([rng.uniform(-0.3, 0.3), rng.uniform(0.25, 0.4), 0.03]) for _ in range(N_SUB)]
        eef = np.array([0.0, 0.1, 0.5]) + rng.uniform(-0.05, 0.05, 3)
        gripR = axis_rot(np.array([1.0, 0, 0]), np.pi) @ axis_rot(np.array([0, 0, 1.0]), rng.uniform(-0.5, 0.5))
        held = False
        hold_off = np.zeros(3)

        def refresh():
            sc.pos, sc.rot, sc.touch, sc.aligned = {}, {}, {}, {}
            sc.pos["griptip_site"] = eef
            sc.rot["grip_site"] = gripR
            sc.pos["table"] = table_pos
            for i in range(N_SUB):
                sc.pos[legs[i]] = leg_pos[i]
                sc.pos[leg_sites[i]] = leg_pos[i] + legR[i] @ np.array([0, 0, -0.1])
                sc.rot[leg_sites[i]] = legR[i] @ axis_rot(np.array([1.0, 0, 0]), np.pi)
                sc.pos["%s_ltgt_site0" % legs[i]] = leg_pos[i] + legR[i] @ np.array([-0.02, 0, 0.01])
                sc.pos["%s_rtgt_site0" % legs[i]] = leg_pos[i] + legR[i] @ np.array([0.02, 0, 0.01])
                sc.pos[table_sites[i]] = table_pos + tableR @ site_off[i]
                sc.rot[table_sites[i]] = tableR
                sc.touch[legs[i]] = (False, False)
                d = np.linalg.norm(sc.pos[table_sites[i]] - sc.pos[leg_sites[i]])
                upc = sc.rot[leg_sites[i]][:, 2] @ tableR[:, 2]
                sc.aligned[leg_sites[i]] = bool(d < 0.03 and upc > 0.97)

        def obs_vec(i):
            o = np.zeros(O_DIM)
            o[0:3] = sc.pos["griptip_site"]
            o[3:6] = sc.pos["%s_ltgt_site0" % legs[i]]
            o[6:9] = sc.pos["%s_rtgt_site0" % legs[i]]
            o[9:12] = sc.pos[legs[i]]
            o[12:15] = sc.pos[leg_sites[i]]
            o[15:18] = sc.pos[table_sites[i]]
            o[18:21] = sc.rot[leg_sites[i]][:, 2]
            o[21:24] = sc.rot[table_sites[i]][:, 2]
            o[24:27] = sc.rot[leg_sites[i]][:, 1]
            o[27:30] = sc.rot[table_sites[i]][:, 1]
            o[30:33] = sc.rot["grip_site"][:, 2]
            o[33:36] = sc.rot["grip_site"][:, 1]
            o[36], o[37] = sc.touch[legs[i]]
            o[38] = sc.aligned[leg_sites[i]]
            return o

        refresh()
        f._reset_reward_variables()
        rec = dict(obs=[], ac=[], connected=[], reward=[], done=[], success=[], phase=[], subtask=[], phase_bonus=[])
        rec0 = np.stack([obs_vec(i) for i in range(N_SUB)])
        attached = [False] * N_SUB
        for t in range(260):
            st = min(f._subtask_step, N_SUB - 1)
            ph = f._phase_i
            leg, lsite, tsite = legs[st], leg_sites[st], table_sites[st]
            grasp = 0.5 * (sc.pos["%s_ltgt_site0" % leg] + sc.pos["%s_rtgt_site0" % leg])
            close = -1.0
            if ph == 0:
                tgt = getattr(f, "_init_eef_pos", eef)
            elif ph == 1:
                tgt = grasp + [0, 0, 0.05]
            elif ph in (2, 3):
                tgt = grasp + [0, 0, -0.015]
                close = 1.0 if ph == 3 else -1.0
            else:
                close = 1.0
                tgt = eef
            k = rng.choice([0.3, 0.6, 1.0])
            if ph < 4:
                eef = eef + k * (np.asarray(tgt) - eef) + rng.normal(0, 0.004, 3)
                # align the gripper's forward axis with the grasp vector
                gv = sc.pos["%s_rtgt_site0" % leg] - sc.pos["%s_ltgt_site0" % leg]
                fw = np.array([gv[0], gv[1], 0.0]) / np.linalg.norm(gv[:2])
                upv = np.array([0, 0, -1.0])
                want = np.stack([np.cross(fw, upv), fw, upv], axis=1)
                if ph >= 1 and rng.rand() < 0.5:
                    gripR = want @ axis_rot(rng.randn(3), rng.uniform(0, 0.1))
            if ph >= 3 and not held and np.linalg.norm(eef - (grasp + [0, 0, -0.015])) < 0.03:
                held = True
                hold_off = leg_pos[st] - eef
            if held:
                if ph == 4:
                    goal = f._lift_leg_pos
                    leg_pos[st] = leg_pos[st] + k * (goal - leg_pos[st]) + rng.normal(0, 0.003, 3)
                elif ph == 5:
                    # rotate the leg so that its site's up matches the table's up
                    want = tableR @ axis_rot(np.array([1.0, 0, 0]), np.pi) @ axis_rot(np.array([0, 0, 1.0]), rng.choice([0, 0.5, 1, 1.5]) * np.pi + rng.uniform(-0.2, 0.2))
                    legR[st] = want if rng.rand() < 0.5 else legR[st] @ axis_rot(rng.randn(3), 0.2)
                    leg_pos[st] = leg_pos[st] + rng.normal(0, 0.003, 3)
                elif ph in (6, 7):
                    site_now = leg_pos[st] + legR[st] @ np.array([0, 0, -0.1])
                    goal = sc.pos[tsite] + ([0, 0, 0.05] if ph == 6 else [0, 0, 0.0])
                    leg_pos[st] = leg_pos[st] + k * (goal - site_now) + rng.normal(0, 0.002, 3)
                    if rng.rand() < 0.3:
                        legR[st] = legR[st] @ axis_rot(np.array([0, 0, 1.0]), rng.uniform(-0.4, 0.4))
                eef = leg_pos[st] - hold_off
            # random disturbances
            if held and rng.rand() < 0.02:
                held = False  # dropped
            if rng.rand() < 0.01:
                table_pos = table_pos + rng.uniform(-0.08, 0.08, 3) * [1, 1, 0]
            refresh()
            if held:
                sc.touch[leg] = (True, True) if rng.rand() < 0.95 else (True, False)
            connected = False
            if held and ph >= 4 and (sc.aligned[lsite] and rng.rand() < 0.5 or rng.rand() < 0.004):
                connected = True
            ac = np.concatenate([rng.uniform(-1, 1, 6), [close * rng.choice([1, 1, 1, -1]) * rng.rand()], [rng.choice([-1.0, 1.0])]])
            f._connected = connected
            obs_all = np.stack([obs_vec(i) for i in range(N_SUB)])
            r, done, info = f._compute_reward(ac)
            rec["obs"].append(obs_all); rec["ac"].append(ac); rec["connected"].append(connected)
            rec["reward"].append(r); rec["done"].append(bool(done)); rec["success"].append(bool(f._success))
            rec["phase"].append(f._phase_i); rec["subtask"].append(f._subtask_step); rec["phase_bonus"].append(info["phase_bonus"])
            if connected and f._subtask_step > st:
                held = False
                attached[st] = True
            f._episode_length += 1
            if done:
                break
        episodes.append((cfg, n_pre, rec0, rec, [s[2] if len(s) == 3 else np.nan for s in site_recipe], grip_init, wz))
    out = {}
    out["n_ep"] = len(episodes)
    for e, (cfg, n_pre, rec0, rec, angles, gi, wz) in enumerate(episodes):
        out["ep%d_flags" % e] = np.array([cfg.diff_rew, cfg.early_termination, cfg.reset_robot_after_attach, n_pre], dtype=np.int32)
        out["ep%d_obs0" % e] = rec0
        for k in rec:
            out["ep%d_%s" % (e, k)] = np.array(rec[k])
    out["angles"] = np.array(episodes[0][4], dtype=float)
    out["has_angles"] = np.array([1, 1, 1, 0])
    out["waypoint_z"] = np.array(episodes[0][6])
    out["grip_init"] = np.array([[np.nan] * 4, [0, 0, 0, 0.37], [0.02, -0.01, 0.03, np.nan], [0, 0, 0, 0.37]])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "dense_reward.npz")
    np.savez_compressed(path, **out)
    ph = np.concatenate([out["ep%d_phase" % e] for e in range(len(episodes))])
    st = np.concatenate([out["ep%d_subtask" % e] for e in range(len(episodes))])
    print("wrote", path, "steps", len(ph), "phase hist", np.bincount(ph, minlength=8), "subtask hist", np.bincount(st),
          "done", sum(out["ep%d_done" % e].any() for e in range(len(episodes))),
          "success", sum(out["ep%d_success" % e].any() for e in range(len(episodes))))


if __name__ == "__main__":
    main()

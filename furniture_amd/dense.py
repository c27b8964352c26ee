"""Host side of the dense-reward env (FurnitureSawyerDenseRewardEnv, furniture_sawyer_dense.py): config defaults
(config/furniture_sawyer_dense.py) and the per-subtask tables the device state machine reads, derived from the furniture's
recipe the way _update_reward_variables does (furniture_sawyer_dense.py:149-216)."""
import numpy as np

# order == the float table uploaded through fsim_set_dense_reward (include/fsim.h, FSIM_DC_*)
DENSE_COEF_DEFAULTS = [
    ("phase_bonus", 5000.0), ("eef_forward_dist_coef", 2.0), ("eef_up_dist_coef", 4.0), ("eef_rot_threshold", 0.95),
    ("gripper_penalty_coef", 1.0), ("move_other_part_penalty_coef", 50.0), ("drop_penalty_coef", 20.0),
    ("early_termination", 0.0), ("init_eef_pos_dist_coef", 100.0), ("move_eef_pos_dist_coef", 100.0),
    ("lower_eef_pos_dist_coef", 1000.0), ("grasp_dist_coef", 200.0), ("lift_z_dist_coef", 500.0), ("lift_xy_dist_coef", 250.0),
    ("lift_z_pos_threshold", 0.02), ("lift_xy_pos_threshold", 0.05), ("align_pos_dist_coef", 100.0),
    ("align_rot_dist_coef", 50.0), ("align_pos_threshold", 0.2), ("align_rot_threshold", 0.85), ("move_pos_dist_coef", 300.0),
    ("move_rot_dist_coef", 50.0), ("move_pos_threshold", 0.06), ("move_rot_threshold", 0.85),
    ("move_fine_pos_exp_coef", -25.0), ("move_fine_pos_dist_coef", 500.0), ("move_fine_rot_dist_coef", 200.0),
    ("aligned_bonus_coef", 10.0), ("ctrl_penalty_coef", 1e-3), ("reset_robot_after_attach", 0.0), ("z_finedist", 0.05),
    ("griptip_site", 0.0), ("grip_site", 0.0), ("phase_ob", 0.0),
]
DENSE_NCOEF = len(DENSE_COEF_DEFAULTS)
# per-subtask row (floats): see FSIM_DS_* in include/fsim.h
DS_LEG_PART, DS_TABLE_PART, DS_LEG_SITE, DS_TABLE_SITE, DS_GL_SITE, DS_GR_SITE, DS_ANGLE, DS_HAS_ANGLES, DS_WAYPOINT_Z, \
    DS_GRIP_INIT_N, DS_GRIP_INIT0, DS_K_LEG, DS_K_TABLE, DS_WORDS = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14, 15, 16


def dense_subtasks(model):
    """One dict per recipe step with the ids the reward needs.  Raises if the furniture has no dense recipe."""
    meta = model.meta
    rec = meta.get("recipe")
    if not rec or "recipe" not in rec:
        raise ValueError("furniture %r has no recipe: the dense-reward env needs one (furniture.py:2036-2044)" % meta.get("furniture_name"))
    sites = list(meta["site_names"])
    parts = list(meta["part_names"])
    conn = [int(s) for s in model.conn_siteid]
    used, out = set(), []
    for st, (leg, table) in enumerate(rec["recipe"]):
        sr = meta["site_recipe"][st]
        leg_site, table_site = sr[0], sr[1]
        angle = float(sr[2]) if len(sr) == 3 else None
        for i in range(len(rec["recipe"])):  # first unused grasp-target site pair of this leg (:196-201)
            gl, gr = "%s_ltgt_site%d" % (leg, i), "%s_rtgt_site%d" % (leg, i)
            if gl not in used and gr not in used:
                used.update((gl, gr))
                break
        gi = rec.get("grip_init_pos")
        gi = gi[st][0] if gi is not None and gi[st] is not None else None
        out.append(dict(leg_part=parts.index(leg), table_part=parts.index(table), leg_site=sites.index(leg_site),
                        table_site=sites.index(table_site), gl_site=sites.index(gl), gr_site=sites.index(gr), angle=angle,
                        has_angles=bool([x for x in leg_site.split(",")[1:-1] if x]),
                        waypoint_z=float(rec["waypoints"][st][0][2]), grip_init=None if gi is None else [float(x) for x in gi],
                        k_leg=conn.index(sites.index(leg_site)), k_table=conn.index(sites.index(table_site))))
    return out, float(rec["z_finedist"]), sites.index("griptip_site"), sites.index("grip_site")


def pack_dense(model, coef=None):
    """-> (coef float32 [DENSE_NCOEF], subtasks float32 [nsub, DS_WORDS]) for fsim_set_dense_reward."""
    subs, zf, griptip, grip = dense_subtasks(model)
    c = dict(DENSE_COEF_DEFAULTS)
    for k, v in (coef or {}).items():
        if k not in c:
            raise KeyError("unknown dense-reward coefficient %r" % k)
        c[k] = float(v)
    c["z_finedist"], c["griptip_site"], c["grip_site"] = zf, float(griptip), float(grip)
    cv = np.array([c[k] for k, _ in DENSE_COEF_DEFAULTS], dtype=np.float32)
    sv = np.zeros((len(subs), DS_WORDS), dtype=np.float32)
    for i, s in enumerate(subs):
        sv[i, :6] = [s["leg_part"], s["table_part"], s["leg_site"], s["table_site"], s["gl_site"], s["gr_site"]]
        sv[i, DS_ANGLE] = np.nan if s["angle"] is None else s["angle"]
        sv[i, DS_HAS_ANGLES] = s["has_angles"]
        sv[i, DS_WAYPOINT_Z] = s["waypoint_z"]
        gi = s["grip_init"]
        sv[i, DS_GRIP_INIT_N] = 0 if gi is None else len(gi)
        if gi is not None:
            sv[i, DS_GRIP_INIT0:DS_GRIP_INIT0 + len(gi)] = gi
        sv[i, DS_K_LEG], sv[i, DS_K_TABLE] = s["k_leg"], s["k_table"]
    return cv, sv

"""CPU restatement (fp64) of the reference's torque-level arm controllers.

TEST INFRASTRUCTURE ONLY (checker for the device controller stage in furniture_amd/csrc/fsim_ctrl.hpp); nothing under
furniture_amd/ imports it.  Pinned to the reference's own classes: tests/golden/controllers.npz was produced by
scripts/make_golden_controllers.py, which runs `furniture/env/controllers/arm_controller.py` itself on injected inputs.

What the env can reach (`FurnitureEnv._load_controller`, furniture.py:1665-1704) is each controller built from
`controllers/controller_config.hjson` with NO overrides, so only that configuration is restated:
linear interpolation, impedance_flag = False, no nullspace posture (initial_joint None), no position / orientation limits.

    ramp: interpolation_steps = floor(ramp_ratio * control_freq / timestep) = floor(0.2 * 20 / 0.002) = 2000
          (arm_controller.py:114 -- the constructor's control_freq default 20, not the env's; the product, not the quotient).
          A goal is therefore approached by 1/2000 of the remaining distance per physics substep and the ramp restarts at
          every policy step: after the 50 substeps of one env step the commanded set-point has moved 2.5 % of the way.

State is a plain dict so that the device code (one float block per arm in the env record) mirrors it field by field.
"""
import numpy as np

# controllers/controller_config.hjson (file:line of each block)
PARAMS = {
    "position_orientation": dict(control_range_pos=0.05, control_range_ori=0.2, kp=150.0, damping=1.0),   # :3-19
    "position": dict(control_range_pos=0.05, kp=150.0, damping=1.0),                                      # :22-37
    "joint_impedance": dict(control_range=[0.2] * 7, kp_max=[100, 100, 100, 100, 50, 30, 10],             # :40-51
                            kp_min=[10, 10, 10, 10, 10, 1, 1], damping_max=[2] * 7, damping_min=[0] * 7),
    "joint_velocity": dict(control_range=[1.0] * 7, kv=[8.0, 7.0, 6.0, 4.0, 2.0, 0.5, 0.1]),              # :54-59
    "joint_torque": dict(control_range=[0.5, 0.5, 0.5, 0.2, 0.2, 0.1, 0.1]),                              # :62-68
}
TYPES = list(PARAMS)
RAMP_RATIO, CTOR_CONTROL_FREQ = 0.2, 20           # arm_controller.py:77, :31
SINGULARITY_THRESHOLD = 0.00025                   # arm_controller.py:783, :788


def control_dim(kind):
    return {"position_orientation": 6, "position": 3}.get(kind, 7)


def control_range(kind):
    p = PARAMS[kind]
    if kind == "position_orientation":
        return np.array([p["control_range_pos"]] * 3 + [p["control_range_ori"]] * 3)
    if kind == "position":
        return np.array([p["control_range_pos"]] * 3)
    return np.array(p["control_range"], dtype=float)


def interpolation_steps(timestep):
    return float(np.floor(RAMP_RATIO * CTOR_CONTROL_FREQ / timestep))


def euler2mat(e):
    """transform_utils.py:360-380."""
    ai, aj, ak = -e[2], -e[1], -e[0]
    si, sj, sk = np.sin(ai), np.sin(aj), np.sin(ak)
    ci, cj, ck = np.cos(ai), np.cos(aj), np.cos(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.array([[cj * ci, cj * si, -sj],
                     [sj * cs - sc, sj * ss + cc, cj * sk],
                     [sj * cc + ss, sj * sc - cs, cj * ck]])


def orientation_error(desired, current):
    """arm_controller.py:180-201: half the sum of the column cross products."""
    return 0.5 * sum(np.cross(current[:, k], desired[:, k]) for k in range(3))


def pinv_thresholded(a):
    """arm_controller.py:781-790: SVD inverse with singular values below 0.00025 zeroed."""
    u, s, vt = np.linalg.svd(a)
    sinv = np.array([0.0 if x < SINGULARITY_THRESHOLD else 1.0 / x for x in s])
    return vt.T @ np.diag(sinv) @ u.T


def new_state(kind):
    """Controller.reset() (arm_controller.py:93-97 and the per-class overrides)."""
    n = control_dim(kind)
    st = dict(kind=kind, step=0)
    if kind in ("position_orientation", "position"):
        st.update(last_goal_position=np.zeros(3), last_goal_orientation=np.eye(3), lin_base=np.zeros(3), lin_delta=np.zeros(3),
                  ori_delta=np.zeros(3), ori_init=np.eye(3), ori_init_live=False, goal_orientation=np.eye(3),
                  goal_orientation_set=False)
    else:
        st.update(last_goal=np.zeros(n), base=np.zeros(n), delta=np.zeros(n))
    return st


def reset_state(st):
    """controller.reset(): what `_reset` calls per arm (furniture.py:1885-1887).  PositionController.reset() does NOT clear
    goal_orientation_set (arm_controller.py:931-932): the orientation captured at the first policy step of the first episode
    is kept for the lifetime of the env."""
    keep = (st.get("goal_orientation"), st.get("goal_orientation_set"))
    fresh = new_state(st["kind"])
    st.clear()
    st.update(fresh)
    if st["kind"] == "position" and keep[1]:
        st["goal_orientation"], st["goal_orientation_set"] = keep


def torques(st, action, policy_step, model, timestep=0.002):
    """update_model + action_to_torques for one physics substep.

    model: dict with the values `update_model` reads (arm_controller.py:109-136): pos (3), mat (3x3), velp, velr (3),
    q, qd (7), Jx, Jr (3x7), M (7x7).  Returns the 7 joint torques (before `+ qfrc_bias`, furniture.py:1756-1758)."""
    kind = st["kind"]
    n = control_dim(kind)
    N = interpolation_steps(timestep)
    rng_ = control_range(kind)
    a = np.clip(np.asarray(action, dtype=float)[:n], -1.0, 1.0) * rng_   # transform_action (:99-107); the ranges are symmetric
    q, qd = np.asarray(model["q"], float), np.asarray(model["qd"], float).copy()
    if kind in ("joint_torque", "joint_velocity"):
        if policy_step:
            st["step"] = 0
            st["base"], st["delta"] = st["last_goal"].copy(), (a - st["last_goal"]) / N   # linear_interpolate (:155-163)
        st["last_goal"] = st["base"] + (st["step"] + 1) * st["delta"]
        if st["step"] < N - 1:
            st["step"] += 1
        if kind == "joint_torque":
            return st["last_goal"].copy()                                                  # :296-299 (no inertia decoupling)
        return np.asarray(PARAMS[kind]["kv"]) * (st["last_goal"] - qd)                    # :364
    if kind == "joint_impedance":
        p = PARAMS[kind]
        kp = (np.asarray(p["kp_max"], float) + np.asarray(p["kp_min"], float)) * 0.5      # :412
        damping = (np.asarray(p["damping_max"], float) + np.asarray(p["damping_min"], float)) * 0.5
        if policy_step:
            st["step"] = 0
            goal = q + a
            if np.linalg.norm(st["last_goal"]) == 0:                                       # :446-447
                st["last_goal"] = q.copy()
            st["base"], st["delta"] = st["last_goal"].copy(), (goal - st["last_goal"]) / N
        st["last_goal"] = st["base"] + (st["step"] + 1) * st["delta"]
        if st["step"] < N - 1:
            st["step"] += 1
        err = st["last_goal"] - q
        kv = 2 * np.sqrt(kp) * damping
        nrm = np.linalg.norm(qd)
        if nrm > 7.0:                                                                      # :485-487 (divides by norm * 7)
            qd = qd / (nrm * 7.0)
        return np.asarray(model["M"], float) @ (kp * err - kv * qd)
    # position / position_orientation
    p = PARAMS[kind]
    pos, R = np.asarray(model["pos"], float), np.asarray(model["mat"], float).reshape(3, 3)
    if policy_step:
        st["step"] = 0
        goal_pos = pos + a[:3]                                                             # set_goal_position (:794-802)
        if kind == "position_orientation":
            st["goal_orientation"] = euler2mat(-a[3:6]).T @ R                              # set_goal_orientation (:808-810)
        elif not st["goal_orientation_set"]:
            st["goal_orientation"], st["goal_orientation_set"] = R.copy(), True            # PositionController (:934-939)
        if np.linalg.norm(st["last_goal_position"]) == 0:
            st["last_goal_position"] = pos.copy()
        # Quirk (arm_controller.py:679-680, :635): on the first policy step after a reset `last_goal_orientation` becomes
        # `self.current_orientation_mat`, which is a VIEW into sim.data.body_xmat (update_model :116), and
        # `orientation_initial_interpolation` aliases it: until the next policy step the "initial" orientation of the ramp
        # follows the hand's current orientation in MuJoCo's memory.
        st["ori_init_live"] = bool((st["last_goal_orientation"] == np.eye(3)).all())
        if st["ori_init_live"]:
            st["last_goal_orientation"] = R.copy()
        st["lin_base"], st["lin_delta"] = st["last_goal_position"].copy(), (goal_pos - st["last_goal_position"]) / N
        st["ori_delta"] = orientation_error(st["goal_orientation"], st["last_goal_orientation"]) / N
        st["ori_init"] = st["last_goal_orientation"].copy()
    st["last_goal_position"] = st["lin_base"] + (st["step"] + 1) * st["lin_delta"]
    god = (st["step"] + 1) * st["ori_delta"]
    if st["ori_init_live"]:
        st["ori_init"] = R.copy()
    st["last_goal_orientation"] = euler2mat(-god).T @ st["ori_init"]
    if st["step"] < N - 1:
        st["step"] += 1
    kp = np.full(6, p["kp"])
    kv = 2 * np.sqrt(kp) * p["damping"]
    f = (st["last_goal_position"] - pos) * kp[:3] - np.asarray(model["velp"], float) * kv[:3]
    t = orientation_error(st["last_goal_orientation"], R) * kp[3:] - np.asarray(model["velr"], float) * kv[3:]
    Jx, Jr, M = np.asarray(model["Jx"], float), np.asarray(model["Jr"], float), np.asarray(model["M"], float)
    Minv = np.linalg.inv(M)
    wrench = np.concatenate([pinv_thresholded(Jx @ Minv @ Jx.T) @ f, pinv_thresholded(Jr @ Minv @ Jr.T) @ t])
    return np.vstack([Jx, Jr]).T @ wrench


def unpack_golden_row(row):
    """Row layout of the *_in arrays of tests/golden/controllers.npz (scripts/make_golden_controllers.py: Scene.record)."""
    o = 0

    def take(k):
        nonlocal o
        v = row[o:o + k]
        o += k
        return v
    return dict(pos=take(3), mat=take(9).reshape(3, 3), velp=take(3), velr=take(3), q=take(7), qd=take(7),
                Jx=take(21).reshape(3, 7), Jr=take(21).reshape(3, 7), M=take(49).reshape(7, 7))


# ---- FurnitureEnv._do_controller_step / _pre_action (furniture.py:3065-3093, 1706-1759), Sawyer -----------------------------
def preprocess_action(action, move_speed):
    """furniture.py:3069-3071: the first three action entries are scaled by move_speed and permuted [-a1, a0, a2] -- for EVERY
    controller type, joint-space ones included (the branch tests the agent, not the controller)."""
    a = np.array(action, dtype=float)
    a[:3] = a[:3] * move_speed
    a[:3] = [-a[1], a[0], a[2]]
    return a


def pre_action_ctrl(st, action, policy_step, model, qfrc_bias_arm, grip_bias, grip_weight, timestep=0.002):
    """One `_pre_action` call: -> ctrl of the 7 arm motors and the 2 finger actuators.
    action: the preprocessed vector [arm command (control_dim), grip, connect]; gripper: format_action 1 -> 2
    (two_finger_gripper.py:66-72) then bias + weight * a from actuator_ctrlrange, unclipped (:1722-1727)."""
    cd = control_dim(st["kind"])
    g = float(action[cd])
    grip = np.asarray(grip_bias, float) + np.asarray(grip_weight, float) * np.array([g, -g])
    tq = torques(st, action[:cd], policy_step, model, timestep)
    return np.asarray(qfrc_bias_arm, float) + tq, grip

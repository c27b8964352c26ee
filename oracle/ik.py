"""CPU statement (fp64) of the batched inverse-kinematics controller that replaces pybullet for control_type "ik" (SURVEY f3).

TEST INFRASTRUCTURE ONLY: the checker of the device IK stage (furniture_amd/csrc/fsim_ik.hpp).

PARITY UNPINNED, by construction: the reference's IK is `pybullet.calculateInverseKinematics` (pybullet==1.9.5, un-vendored
binary; call sites controllers/sawyer_ik_controller.py:194-217, 20 calls per control step :263-267) on the 7-dof arm, which
is redundant for a 6-d pose target -- the joint solution it returns depends on Bullet's internal damped-least-squares /
null-space iteration and cannot be reproduced without its source.  What IS restated exactly, around that solver:
  * the chain it solves on (the same URDF, parsed into the compiled model: furniture_amd/mjcf/urdf_chain.py) -- checked
    against the MJCF kinematics in tests/test_ik.py;
  * the target bookkeeping: ik_robot_target_pos += dpos * user_sensitivity(0.3) in the robot base frame
    (sawyer_ik_controller.py:240-246), orientation target = commanded hand orientation . Rz(-90 deg) (:248-258), both for the
    centre-of-mass frame of link right_l6 (getLinkState()[0:2], :162-163);
  * the P controller joint error -> velocity: v = clip(-5 (q - q_cmd), -1, 1) (:75-84);
  * the env side (furniture.py:2899-2991): action scaling / permutation, _bounded_d_pos, the accumulated commanded orientation
    `_initial_right_hand_quat`, three closed-loop repeats of `_do_simulation`.
The solver itself is a plain damped-least-squares iteration with a null-space pull towards the reference's rest pose,
a FIXED iteration count (so the fp32 device run and this fp64 run take the same path), joint-limit clamping as the reference
passes to Bullet (:207-209).  Validation is by property (the solution reaches the target) and by task behaviour.
"""
import numpy as np

USER_SENSITIVITY = 0.3          # sawyer_ik_controller.py:47
IK_ITERS = 12                   # fixed; Bullet: up to 20 calls x 20 internal iterations with a 1e-4 residual threshold
IK_TAIL = 4                     # the last iterations are pure task-space steps (the damped projector leaks O(lambda^2) into the task)
IK_DAMPING = 0.05               # lambda of (J J' + lambda^2 I)
IK_NULL_GAIN = 0.01             # pull towards rest_poses inside the null space of J
IK_LOWER = np.array([-3.05, -3.82, -3.05, -3.05, -2.98, -2.98, -4.71])   # :207
IK_UPPER = np.array([3.05, 2.28, 3.05, 3.05, 2.98, 2.98, 4.71])          # :208


def qmul(a, b):  # wxyz
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def q2m(q):  # wxyz -> 3x3
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def fk(model, q, arm=0):
    """Pose of the end-effector frame (CoM frame of the IK chain's end link: right_l6 for Sawyer, right / left_gripper for Baxter) in
    the robot base frame, plus joint origins and axes.  URDF joint: child = parent . Trans(xyz) . Rot(rpy) . Rz(q_i); fixed joints
    are folded into the following joint's origin (furniture_amd/mjcf/urdf_chain.py)."""
    R, p = np.eye(3), np.zeros(3)
    origins, axes = [], []
    jp, jq = model.ik_joint_pos[7 * arm:7 * arm + 7], model.ik_joint_quat[7 * arm:7 * arm + 7]
    for i in range(7):
        p = p + R @ jp[i]
        R = R @ q2m(jq[i])
        origins.append(p.copy())
        axes.append(R[:, 2].copy())
        c, s = np.cos(q[i]), np.sin(q[i])
        R = R @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    eq = np.asarray(model.ik_eef_quat, float).reshape(-1, 4)[arm]
    return p + R @ np.asarray(model.ik_eef_pos, float).reshape(-1, 3)[arm], R @ q2m(eq), np.array(origins), np.array(axes)


def rotvec(R):
    """axis * angle of a rotation matrix (small-angle safe)."""
    v = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(v)
    c = 0.5 * (np.trace(R) - 1.0)
    if s < 1e-12:
        return v
    return v * (np.arctan2(s, c) / s)


def solve(model, q0, target_pos, target_R, iters=IK_ITERS, arm=0, rest=None):
    """joint_positions_for_eef_command's inner solve: from the current joints to the target pose (both in the base frame).
    rest: the rest pose handed to the solver (Sawyer: a fixed pose; Baxter: the current joints, baxter_ik_controller.py:321)."""
    q = np.array(q0, dtype=float)
    rest = np.asarray(model.ik_rest, float).reshape(-1, 7)[arm] if rest is None else np.asarray(rest, float)
    lower = np.asarray(getattr(model, "ik_lower", IK_LOWER), float).reshape(-1, 7)[arm]
    upper = np.asarray(getattr(model, "ik_upper", IK_UPPER), float).reshape(-1, 7)[arm]
    for k in range(iters):
        p, R, o, z = fk(model, q, arm)
        e = np.concatenate([target_pos - p, rotvec(target_R @ R.T)])
        J = np.zeros((6, 7))
        for i in range(7):
            J[:3, i] = np.cross(z[i], p - o[i])
            J[3:, i] = z[i]
        A = J @ J.T + IK_DAMPING ** 2 * np.eye(6)
        y = np.linalg.solve(A, e)
        dq = J.T @ y
        if k < iters - IK_TAIL:
            n = IK_NULL_GAIN * (rest - q)
            dq = dq + n - J.T @ np.linalg.solve(A, J @ n)
        q = np.clip(q + dq, lower, upper)
    return q


def rot_z(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def velocities(q, q_cmd, gain=5.0):
    """get_control's P controller (sawyer_ik_controller.py:75-84: -5 delta; baxter_ik_controller.py:86-95: -2 delta), clipped to +-1."""
    return np.clip(-gain * (np.asarray(q) - np.asarray(q_cmd)), -1.0, 1.0)


def arm_action_slices(agent, control_type, arm):
    """Where an arm's command sits in the env action (furniture.py:2911-2958, 2994-3018): -> (dpos slice, rotation slice, grip index
    from the END of the action).  Sawyer: [dpos 3, rot 3 | quat 4, grip, connect]; Baxter: [right dpos, rot, left dpos, rot, grip_r,
    grip_l, connect]."""
    nrot = 3 if control_type == "ik" else 4
    narm = 1 if agent == "Sawyer" else 2
    o = arm * (3 + nrot)
    return slice(o, o + 3), slice(o + 3, o + 3 + nrot), -(1 + narm) + arm


def preprocess(control_type, action, move_speed, rotate_speed, hand_pos, right_hand_quat, initial_quat, agent="Sawyer", arm=0):
    """FurnitureEnv._do_ik_step up to the controller call (furniture.py:2911-2958 for "ik", :2999-3027 for "ik_quaternion") and
    _make_input (:1332-1343), for one arm: -> (d_pos, rotation 3x3, new _initial_<arm>_hand_quat, gripper action).  The reference
    hands the xyzw quaternion _initial_right_hand_quat to euler_to_quat, whose pyquaternion reads it as wxyz: the same functions are
    used here in the same way (pinned by tests/golden/controllers.npz, ikstep_*)."""
    from furniture_amd import transform_utils as T
    action = np.array(action, dtype=float)
    sp, sr, ig = arm_action_slices(agent, control_type, arm)
    dp = action[sp] * move_speed
    dp = np.array([-dp[1], dp[0], dp[2]])
    hand_pos = np.asarray(hand_pos, float)
    d_pos = np.clip(dp, np.array([-1.5, -1.5, 0.0]) - hand_pos, np.array([1.5, 1.5, 1.5]) - hand_pos)  # :170-171, 1252-1258
    if control_type == "ik_quaternion":
        d_quat = T.convert_quat(action[sr])
        new_initial = np.asarray(initial_quat, float)
    else:
        new_initial = np.array(T.euler_to_quat(action[sr] * rotate_speed, initial_quat))
        d_quat = T.quat_multiply(T.quat_inverse(right_hand_quat), new_initial)
    rotation = T.quat2mat(T.quat_multiply(right_hand_quat, d_quat))
    return d_pos, rotation, new_initial, action[ig]

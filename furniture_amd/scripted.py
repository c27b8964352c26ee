"""Scripted pick-and-attach policy for Sawyer + table_lack_0825 under ``control_type="ik_quaternion"`` -- a scenario generator
in the spirit of the reference's ``furniture/env/furniture_sawyer_gen.py`` (phase-based scripted demonstrations), reduced to the
first assembly subtask and written against the observation dict only (+ the connector-site tables of the compiled model):

  orient (gripper down, fingers across leg 0) -> above the leg -> descend -> close -> lift -> quarter turn about the finger axis
  (the leg hangs vertically, connector down) -> carry over the nearest table connector -> connect

It is open loop per phase (fixed step counts, proportional set-points), so some placements fail; on the device 7 of 8 random
placements complete the subtask (tests/test_ik.py).  Works on anything with the batched interface:
``step(actions [n, 9]) -> (ob dict, reward, done, info)`` with ``ob["object_ob"] [n, 7 * parts]`` and ``ob["robot_ob"] [n, 15]``.
"""

import numpy as np

GRIPPER_DOWN = np.array([[-1.0, 0, 0], [0, 1.0, 0], [0, 0, -1.0]])       # hand z -> world -z, finger axis (hand y) -> world y
QUARTER_TURN_Y = np.array([[0, 0, 1.0], [0, 1.0, 0], [-1.0, 0, 0]])      # +90 deg about world y: the leg's +x end -> down


def _rot(q_wxyz):
    """[..., 4] wxyz -> [..., 3, 3] (batched)"""
    q = np.asarray(q_wxyz, dtype=float)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - w * z); R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y); R[..., 2, 1] = 2 * (y * z + w * x); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


class PickAndAttach:
    """policy = PickAndAttach(model, n); for a in policy.actions(ob): ob, ... = env.step(a); policy.observe(ob)."""

    def __init__(self, model, n, leg=0, table=4, leg_conn=0, table_conn=6, attach=True):
        self.m, self.n, self.attach = model, n, attach
        self.leg, self.table = leg, table
        self.s_leg, self.s_tab = int(model.conn_siteid[leg_conn]), int(model.conn_siteid[table_conn])
        assert int(model.site_bodyid[self.s_leg]) == int(model.part_bodyid[leg]) and int(model.site_bodyid[self.s_tab]) == int(model.part_bodyid[table])

    def _action(self, obj, rob, target_R, target, grip, connect, maxrot):
        """Vectorised over the batch: orientation servo in the hand frame + proportional position set-point."""
        n = self.n
        a = np.zeros((n, 9), dtype=np.float32)
        R = _rot(rob[:, [8, 5, 6, 7]])                                     # eef_quat is xyzw (furniture_sawyer.py:138-140)
        E = np.einsum("nji,jk->nik", R, target_R)                          # R^T . target: remaining rotation, in the hand frame
        w = 0.5 * np.stack([E[:, 2, 1] - E[:, 1, 2], E[:, 0, 2] - E[:, 2, 0], E[:, 1, 0] - E[:, 0, 1]], axis=1)
        nw = np.linalg.norm(w, axis=1)
        ang = np.arcsin(np.minimum(1.0, nw))
        far = (np.trace(E, axis1=1, axis2=2) < 1.0) & (nw < 0.9)
        ang = np.where(far, np.pi - ang, ang)
        th = np.minimum(ang, maxrot)
        a[:, 3] = np.cos(th / 2)                                           # quaternion (wxyz) relative to the hand
        a[:, 4:7] = w / (nw[:, None] + 1e-12) * np.sin(th / 2)[:, None]
        if target is not None:
            a[:, :3] = np.clip((target(obj, rob) - rob[:, 2:5]) / 0.03 * 0.5, -1, 1)  # 3 cm of target per unit action
        a[:, 7], a[:, 8] = grip, connect
        return a

    def run(self, step, ob):
        """Drive the batched env through the script.  step(actions) -> (ob, reward, done, info).  Returns
        (summed reward [n], max num_connected [n], final ob)."""
        lo, m = 7 * self.leg, self.m
        obj, rob = _to_numpy(ob["object_ob"]), _to_numpy(ob["robot_ob"])
        leg0 = obj[:, lo:lo + 3].copy()
        total, ncon = np.zeros(self.n), np.zeros(self.n, dtype=int)
        state = {"ob": ob}

        def phase(steps, target_R, target, grip, connect=-1.0, maxrot=0.15):
            nonlocal total, ncon
            for _ in range(steps):
                obj, rob = _to_numpy(state["ob"]["object_ob"]), _to_numpy(state["ob"]["robot_ob"])
                ob2, rew, done, info = step(self._action(obj, rob, target_R, target, grip, connect, maxrot))
                state["ob"] = ob2
                total += _to_numpy(rew).reshape(self.n)
                ncon = np.maximum(ncon, _to_numpy(info["num_connected"]).reshape(self.n).astype(int))
        at = lambda z: (lambda obj, rob: np.concatenate([leg0[:, :2], np.full((self.n, 1), z)], axis=1))
        phase(25, GRIPPER_DOWN, None, -1.0)
        phase(30, GRIPPER_DOWN, at(0.12), -1.0)
        phase(30, GRIPPER_DOWN, at(0.028), -1.0)       # finger tips straddle the 3 cm leg
        phase(8, GRIPPER_DOWN, at(0.028), 1.0)         # close
        if not self.attach:
            phase(30, GRIPPER_DOWN, at(0.15), 1.0)
            return total, ncon, state["ob"]
        phase(30, GRIPPER_DOWN, at(0.25), 1.0)         # high enough for the 26 cm leg to hang vertically
        turned = QUARTER_TURN_Y @ GRIPPER_DOWN
        phase(40, turned, at(0.25), 1.0, maxrot=0.08)
        to = 7 * self.table
        tab = obj[:, to:to + 7].copy()                 # the table top has not moved
        tab_conn = tab[:, :3] + _rot(tab[:, 3:7]) @ m.site_pos[self.s_tab]

        def over_table(obj, rob):
            leg_conn = obj[:, lo:lo + 3] + _rot(obj[:, lo + 3:lo + 7]) @ m.site_pos[self.s_leg]
            return tab_conn + np.array([0, 0, 0.03]) + (rob[:, 2:5] - leg_conn)
        phase(50, turned, over_table, 1.0, maxrot=0.08)
        phase(5, turned, over_table, 1.0, connect=1.0, maxrot=0.08)   # connect > 0 while both fingers hold the leg
        return total, ncon, state["ob"]

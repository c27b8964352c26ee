"""Scripted assembly policy for Sawyer + table_lack_0825 under ``control_type="ik_quaternion"`` -- a scenario generator in the
spirit of the reference's ``furniture/env/furniture_sawyer_gen.py`` (phase-based scripted demonstrations), written against the
observation dict only (+ the connector-site tables of the compiled model).  Per leg:

  orient (gripper down, fingers across the leg) -> above the leg -> descend -> close -> lift -> quarter turn about the finger axis
  (the leg hangs vertically, connector down) -> carry over the nearest free table connector -> connect -> release, back off

``run(legs=(0, 3, 1, 2))`` assembles the whole table (num_connected 4 = success).  Phases are closed loop on the end-effector
position (a phase ends when every env of the batch is within tolerance of its set-point, with a step cap) and each env asks to connect
on its own schedule; there is no re-grasp, so some placements fail a leg (tests/test_ik.py states the measured rates).  Works on
anything with the batched interface: ``step(actions [n, 9]) -> (ob dict, reward, done, info)`` with ``ob["object_ob"] [n, 7 * parts]``
and ``ob["robot_ob"] [n, 15]``.
"""

import numpy as np

GRIPPER_DOWN = np.array([[-1.0, 0, 0], [0, 1.0, 0], [0, 0, -1.0]])       # hand z -> world -z, finger axis (hand y) -> world y
FULL_TABLE = (0, 3, 1, 2)                                                # near pair first, then the far pair
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

    def __init__(self, model, n, leg=0, table=4, leg_conn=0, table_conn=6, attach=True, gain=0.3, hover=0.03):
        self.m, self.n, self.attach, self.gain, self.hover = model, n, attach, gain, hover   # hover: connector gap at which connect is sent
        self.leg, self.table = leg, table
        self.s_leg, self.s_tab = int(model.conn_siteid[leg_conn]), int(model.conn_siteid[table_conn])
        assert int(model.site_bodyid[self.s_leg]) == int(model.part_bodyid[leg]) and int(model.site_bodyid[self.s_tab]) == int(model.part_bodyid[table])

    def _action(self, obj, rob, target_R, target, grip, connect, maxrot, speed=1.0):
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
            a[:, :3] = np.clip((target(obj, rob) - rob[:, 2:5]) / 0.03 * self.gain, -speed, speed)  # 3 cm of target per unit action
        a[:, 7], a[:, 8] = grip, connect
        return a

    def run(self, step, ob, legs=None, table_connectors=None):
        """Drive the batched env through the script.  step(actions) -> (ob, reward, done, info).  legs: which legs to attach, in order
        (default: leg 0 only; FULL_TABLE = (0, 3, 1, 2) takes the two legs next to the robot first, then the far pair).  Each leg goes
        to the free connector of the table top nearest to it, or -- table_connectors: one connector index (the model's conn_* tables) per
        leg -- to the one a recipe prescribes (the dense-reward env ends the episode on any other, furniture_sawyer_dense.py).  Returns (summed reward [n], max num_connected [n], final ob)."""
        m, n = self.m, self.n
        legs = [self.leg] if legs is None else list(legs)
        total, ncon = np.zeros(n), np.zeros(n, dtype=int)
        state = {"ob": ob}
        to = 7 * self.table
        table_conns = [k for k in range(len(m.conn_siteid)) if int(m.conn_partid[k]) == self.table]
        used = np.zeros((n, len(table_conns)), dtype=bool)

        def phase(steps, target_R, target, grip, connect=-1.0, maxrot=0.15, tol=None, speed=1.0, connect_within=None):
            """`steps` env steps, or -- with tol -- until every env's end effector is within tol of its set-point (at most 3 x steps).
            connect_within: every env asks to connect on the steps on which it is within that distance of its set-point, and the phase
            ends when all envs have one more connection than at its start."""
            nonlocal total, ncon
            n0 = ncon.copy()
            for k in range(steps if tol is None and connect_within is None else 3 * steps):
                obj, rob = _to_numpy(state["ob"]["object_ob"]), _to_numpy(state["ob"]["robot_ob"])
                err = np.abs(target(obj, rob) - rob[:, 2:5]).max(axis=1) if target is not None else np.zeros(n)
                if tol is not None and k >= 3 and err.max() < tol:
                    break
                if connect_within is not None:
                    if (ncon > n0).all():
                        break
                    connect = np.where((err < connect_within) & (ncon == n0), 1.0, -1.0)
                ob2, rew, done, info = step(self._action(obj, rob, target_R, target, grip, connect, maxrot, speed))
                state["ob"] = ob2
                total += _to_numpy(rew).reshape(n)
                ncon = np.maximum(ncon, _to_numpy(info["num_connected"]).reshape(n).astype(int))
        turned = QUARTER_TURN_Y @ GRIPPER_DOWN
        for li, leg in enumerate(legs):
            lo = 7 * leg
            s_leg = int(m.conn_siteid[[k for k in range(len(m.conn_siteid)) if int(m.conn_partid[k]) == leg][0]])
            obj, rob = _to_numpy(state["ob"]["object_ob"]), _to_numpy(state["ob"]["robot_ob"])
            leg0 = obj[:, lo:lo + 3].copy()
            at = lambda z, leg0=leg0: (lambda obj, rob: np.concatenate([leg0[:, :2], np.full((n, 1), z)], axis=1))
            if li:   # coming from the previous attach: release, back off, go up, turn the gripper down again
                phase(10, turned, None, -1.0)
                here = _to_numpy(state["ob"]["robot_ob"])[:, 2:5].copy()
                phase(20, turned, lambda obj, rob: here + np.array([0.12, 0.0, 0.08]), -1.0, tol=0.01)
                phase(30, GRIPPER_DOWN, lambda obj, rob: np.concatenate([here[:, :2] + np.array([0.12, 0.0]), np.full((n, 1), 0.45)], axis=1), -1.0, tol=0.01)
                phase(45, GRIPPER_DOWN, at(0.45), -1.0, tol=0.008)
            else:
                phase(25, GRIPPER_DOWN, None, -1.0)
            phase(30, GRIPPER_DOWN, at(0.12), -1.0, tol=0.003)
            phase(12, GRIPPER_DOWN, at(0.034), -1.0, tol=0.003, speed=0.4)   # finger tips straddle the 3 cm leg, just off the floor
            phase(8, GRIPPER_DOWN, at(0.034), 1.0)         # close
            if not self.attach:
                phase(30, GRIPPER_DOWN, at(0.15), 1.0)
                return total, ncon, state["ob"]
            high = 0.25 if len(legs) == 1 else 0.5         # with a leg already standing on the table, carry above it
            phase(30 if high < 0.3 else 40, GRIPPER_DOWN, at(high), 1.0, tol=0.01)
            phase(40 if high < 0.3 else 45, turned, at(high), 1.0, maxrot=0.08)
            obj = _to_numpy(state["ob"]["object_ob"])
            tab = obj[:, to:to + 7].copy()                 # where the table top is now
            conn_pos = np.stack([tab[:, :3] + _rot(tab[:, 3:7]) @ m.site_pos[int(m.conn_siteid[k])] for k in table_conns], axis=1)  # [n, 4, 3]
            dist = np.linalg.norm(conn_pos - leg0[:, None, :], axis=2) + 1e3 * used
            pick = dist.argmin(axis=1) if table_connectors is None else np.full(n, table_conns.index(int(table_connectors[li])))
            used[np.arange(n), pick] = True
            tab_conn = conn_pos[np.arange(n), pick]

            def over_table(dz):
                def f(obj, rob):
                    leg_conn = obj[:, lo:lo + 3] + _rot(obj[:, lo + 3:lo + 7]) @ m.site_pos[s_leg]
                    return tab_conn + np.array([0, 0, dz]) + (rob[:, 2:5] - leg_conn)
                return f
            if high > 0.3:
                phase(60, turned, over_table(0.30), 1.0, maxrot=0.08, tol=0.01)
            phase(50, turned, over_table(self.hover), 1.0, maxrot=0.08, tol=0.02)
            phase(20, turned, over_table(self.hover), 1.0, maxrot=0.08, connect_within=0.008)   # connect > 0 while both fingers hold the leg
        return total, ncon, state["ob"]

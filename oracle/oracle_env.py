"""Single-env CPU restatement of FurnitureEnv's step()/reset() control flow over the CPU oracle.

TEST INFRASTRUCTURE (see oracle/fsim_oracle.h): this is the behavioural spec the device env logic
(furniture_amd/csrc/fsim_env.hpp) is diffed against, and BASELINE config 1 ("FurnitureCursorEnv +
toy_table, 1 env, CPU reference step(), plumbing, no GPU").  Each method cites the reference lines
it follows; ``F.py`` = /root/reference/furniture/env/furniture.py.

PARITY UNPINNED against MuJoCo itself (the physics under ``self.sim`` is oracle/fsim_oracle.c).  The env LOGIC here is
pinned where the reference's own methods can be run without MuJoCo: _is_aligned, _find_group/_merge_groups,
_compute_reward, _try_connect, _connect/_activate_weld/_get_next_subtask, _setup_action, _get_obs and _step_discrete
reproduce the outputs of FurnitureEnv's methods called on a fake ``self``
(tests/golden/env_logic.npz, scripts/make_golden_env_logic.py, tests/test_env_logic_golden.py).
"""

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from furniture_amd import transform_utils as T  # noqa: E402
from furniture_amd.transform_utils import Quaternion  # noqa: E402
from oracle.oracle_sim import OracleSim, SimUnstable  # noqa: E402


class OracleConfig:
    """Defaults of furniture/config/furniture.py (+ dense-env overrides where noted)."""

    def __init__(self, **kw):
        self.control_freq = 10
        self.max_episode_steps = 2000
        self.discrete_grip = True
        self.rescale_actions = True
        self.auto_align = True
        self.furn_xyz_rand = 0.02
        self.furn_rot_rand = 3
        self.agent_xyz_rand = 0.001
        self.alignment_pos_dist = 0.1
        self.alignment_rot_dist_up = 0.9
        self.alignment_rot_dist_forward = 0.9
        self.alignment_project_dist = 0.3
        self.ctrl_penalty_coef = 1e-3
        self.unstable_penalty_coef = 100
        self.success_reward = 100
        self.touch_reward = 10
        self.pick_reward = 100
        self.move_speed = 0.1
        self.rotate_speed = 22.5
        self.cursor_boundary = 1.5
        self.seed = 123
        self.solver = "newton"
        self.solver_iterations = 100
        self.solver_tolerance = 1e-8
        self.dense = None  # an oracle.dense_reward.DenseConfig -> FurnitureSawyerDenseRewardEnv behaviour
        self.preassembled = []   # config.preassembled (furniture.py:163): weld ids / recipe steps every reset starts from
        self.num_connects = None  # config.num_connects
        self.assembled = False    # config.assembled (furniture.py:1502-1503, 1526-1530)
        self.fix_init = False     # config.fix_init (furniture.py:1518-1525)
        self.control_type = "impedance"  # or one of NEW_CONTROLLERS (furniture.py:41-47): needs the __torque compiled model
        for k, v in kw.items():
            setattr(self, k, v)


def sample_placement(rng, model, cfg):
    """UniformRandomSampler.sample() with every part preset from the XML (tasks/placement_sampler.py:138-190,
    sample_quat :124-136).  Returns (pos dict by part index, quat dict)."""
    n = model.nparts
    init = model.part_initqpos
    r = cfg.furn_xyz_rand
    placed = []
    pos_arr, quat_arr = {}, {}
    for i in range(n):
        obj_r = model.part_hradius[i]
        ok = False
        for _ in range(10000):
            x = init[i, 0] + rng.uniform(high=max(-r, r), low=min(-r, r))
            y = init[i, 1] + rng.uniform(high=max(-r, r), low=min(-r, r))
            z = init[i, 2] + 0.01
            valid = True
            for (px, py, pr) in placed:
                if np.linalg.norm([x - px, y - py], 2) <= pr + obj_r:
                    valid = False
                    break
            if valid:
                rot_range = [-cfg.furn_rot_rand, cfg.furn_rot_rand]
                noise = rng.uniform(high=max(rot_range), low=max(rot_range))  # quirk Q2: constant, RNG still advances
                quat = T.euler_to_quat([noise, 0, 0], Quaternion(init[i, 3:7]))
                placed.append((x, y, obj_r))
                pos_arr[i] = np.array([x, y, z])
                quat_arr[i] = np.array(quat)
                ok = True
                break
        if not ok:
            raise RuntimeError("Cannot place all objects on the desk")
    return pos_arr, quat_arr


class FurnitureEnvOracle:
    def __init__(self, model, config=None):
        self.m = model
        self.cfg = config or OracleConfig()
        self.agent = model.meta["agent"]
        self.sim = OracleSim(model)
        self.sim.set_solver(self.cfg.solver_iterations, self.cfg.solver_tolerance, self.cfg.solver)
        self._rng = np.random.RandomState(self.cfg.seed)
        self.nparts = model.nparts
        self.arms = model.meta["arms"]
        self._num_connect_steps = 10 if self.agent == "Cursor" else 0
        self._gravity_compensation = 1 if self.agent == "Cursor" else 0
        self._n_substeps = int((1.0 / self.cfg.control_freq) / model.opt[0])
        self._has_recipe = bool(model.meta.get("has_recipe"))
        self.init_pos = None
        self.init_quat = None
        self.reset_draws = None  # filled by reset(): what a device reset table must contain
        self.attach_draws = []   # config.reset_robot_after_attach: the joint-noise draw of every _connect so far (what the device's attach table held)
        self._fail = False
        self._dense = None
        self._ctrl = None
        self._ik = self.cfg.control_type in ("ik", "ik_quaternion")
        if self._ik:
            from oracle import ik as IK
            assert self.agent in ("Sawyer", "Baxter")
            self._IK = IK
            self._action_repeat = 3  # furniture.py:172
        elif self.cfg.control_type != "impedance":
            # _load_controller (F.py:1665-1704) builds one controller per arm at construction; controller.reset() runs only in
            # _reset_internal (F.py:1885-1887), i.e. on the FIRST reset -- the controller state survives later resets.
            from oracle import controllers as C
            assert self.cfg.control_type in C.TYPES and self.agent == "Sawyer", self.cfg.control_type
            self._C = C
            self._ctrl = [C.new_state(self.cfg.control_type) for _ in self.arms]
        if self.cfg.dense is not None:
            from furniture_amd.dense import dense_subtasks
            from oracle.dense_reward import DenseReward
            self._dsub, zf, self._griptip_site, self._grip_site = dense_subtasks(model)
            self._dense = DenseReward(self.cfg.dense, self._dsub, zf, self.nparts - 1, 0)

    def _dense_obs(self, st):
        """observables of subtask st (oracle/dense_reward.py O_*) from the current sim state."""
        from oracle.dense_reward import O_DIM
        d, s = self.sim.data, self._dsub[st]
        R = lambda site: d.site_xmat[site].reshape(3, 3)
        o = np.zeros(O_DIM)
        o[0:3] = d.site_xpos[self._griptip_site]
        o[3:6], o[6:9] = d.site_xpos[s["gl_site"]], d.site_xpos[s["gr_site"]]
        o[9:12] = d.xpos[self.m.part_bodyid[s["leg_part"]]]
        o[12:15], o[15:18] = d.site_xpos[s["leg_site"]], d.site_xpos[s["table_site"]]
        o[18:21], o[21:24] = R(s["leg_site"])[:, 2], R(s["table_site"])[:, 2]
        o[24:27], o[27:30] = R(s["leg_site"])[:, 1], R(s["table_site"])[:, 1]
        o[30:33], o[33:36] = R(self._grip_site)[:, 2], R(self._grip_site)[:, 1]
        L, Rr, _ = self._touch_sets()[0]
        o[36], o[37] = s["leg_part"] in L, s["leg_part"] in Rr
        return o

    # ---- small accessors (F.py:3107-3310) --------------------------------------------------
    def _part_qpos(self, i):
        a = self.m.part_qposadr[i]
        return self.sim.data.qpos[a:a + 7].copy()

    def _set_part_qpos(self, i, pos, rot):
        a = self.m.part_qposadr[i]
        self.sim.data.qpos[a:a + 3] = pos
        self.sim.data.qpos[a + 3:a + 7] = rot

    def _site_xpos_xquat(self, site):
        b = self.m.site_bodyid[site]
        q = Quaternion(self.sim.data.xquat[b]) * Quaternion(self.m.site_quat[site])
        return np.hstack([self.sim.data.site_xpos[site], list(q)])

    def _find_group(self, i):
        if self._group[i] == i:
            return i
        self._group[i] = self._find_group(self._group[i])
        return self._group[i]

    def _merge_groups(self, i, j):
        self._group[self._find_group(i)] = self._find_group(j)

    def _stop_object(self, i, gravity=1):
        b = self.m.part_bodyid[i]
        self.sim.data.xfrc_applied[b] = [0, 0, -gravity * self.m.opt[3] * self.m.body_mass[b], 0, 0, 0]
        d = self.m.part_dofadr[i]
        self.sim.data.qvel[d:d + 6] = 0
        self.sim.data.qfrc_applied[d:d + 6] = 0

    def _slow_object(self, i):
        b = self.m.part_bodyid[i]
        self.sim.data.xfrc_applied[b] = [0, 0, -self.m.opt[3] * self.m.body_mass[b], 0, 0, 0]
        d = self.m.part_dofadr[i]
        self.sim.data.qvel[d:d + 6] = np.clip(self.sim.data.qvel[d:d + 6], -0.2, 0.2)
        self.sim.data.qfrc_applied[d:d + 6] = 0

    def _robot_dofs(self):
        return np.concatenate([self.m.arm_dofadr, self.m.grip_dofadr]).astype(int)

    def _gravity_comp(self):
        rd = self._robot_dofs()
        self.sim.data.qfrc_applied[rd] = self.sim.data.qfrc_bias[rd]

    def _fs(self):
        self.sim.forward()
        self.sim.step()

    # ---- reset (F.py:1406-1663) ------------------------------------------------------------
    def _initialize_robot_pos(self, record=None):
        if self.agent != "Cursor":
            noise = self._rng.uniform(low=-self.cfg.agent_xyz_rand, high=self.cfg.agent_xyz_rand, size=self.m.arm_initqpos.shape)
            (self.reset_draws["noise"] if record is None else record).append(noise.copy())
            self.sim.data.qpos[self.m.arm_qposadr] = self.m.arm_initqpos + noise
            self.sim.data.qpos[self.m.grip_qposadr] = self.m.grip_initqpos
        else:
            for k, x in enumerate((-0.2, 0.2)):
                self.sim.model.body_pos[self.m.cursor_bodyid[k]] = [x, 0.0, self.cfg.move_speed / 2]

    def set_init_qpos(self, init_qpos):
        """F.py:315-316: {qpos, qvel} (get_env_state's format) or None"""
        self._init_qpos = init_qpos

    def set_subtask(self, subtask, num_connects=None):
        """F.py:204-207"""
        self.cfg.preassembled = list(range(subtask))
        self.cfg.num_connects = num_connects

    def _project_connector_quat(self, k1, k2, angle=None):
        """F.py:1201-1222: connector k2's xquat when aligned with connector k1 (at `angle` degrees about k1's up axis)"""
        m = self.m
        R1 = self.sim.data.site_xmat[m.conn_siteid[k1]].reshape(3, 3)
        R2 = self.sim.data.site_xmat[m.conn_siteid[k2]].reshape(3, 3)
        up1, f1, f2 = R1[:, 2].copy(), R1[:, 1].copy(), R2[:, 1].copy()
        if angle is None:
            cs = T.cos_siml(f1, f2)
            rp = T.rotate_vector_cos_siml(f1, up1, cs, 1)
            rn = T.rotate_vector_cos_siml(f1, up1, cs, -1)
            fr = rp if T.cos_siml(rp, f2) > T.cos_siml(rn, f2) else rn
        else:
            fr = T.rotate_vector(f1, up1, angle)
        return T.convert_quat(T.lookat_to_quat(up1, fr), "wxyz")

    def reset(self):
        m, sim = self.m, self.sim
        self.reset_draws = {"noise": []}
        sim.reset()
        ct0, ca0 = m.geom_contype.copy(), m.geom_conaffinity.copy()
        robot = m.geom_is_robot.astype(bool)
        sim.model.geom_contype[:] = ct0
        sim.model.geom_conaffinity[:] = ca0
        sim.model.geom_contype[robot] = 0
        sim.model.geom_conaffinity[robot] = 0
        pc = m.geom_is_partcol.astype(bool)
        sim.model.geom_contype[pc] = 1
        sim.model.geom_conaffinity[pc] = 1
        self._group = list(range(self.nparts))
        self._connect_step = 0
        self._connected = False
        self._connected_sites = set()
        self._connected_body1 = None
        self._num_connected = 0
        self._prev_num_connected = 0
        self._site1_id = self._site2_id = -1
        if self.agent == "Cursor":
            self._cursor_selected = [None, None]
        pre = list(getattr(self.cfg, "preassembled", None) or [])
        nc = getattr(self.cfg, "num_connects", None)
        self._success_num_conn = self.nparts - 1 if nc is None else nc + len(pre)  # F.py:1476-1481
        self._touched = [False] * self.nparts
        self._picked = [False] * self.nparts
        sim.model.eq_active[:] = 0
        sim.model.eq_data[:] = m.eq_data0
        if getattr(self.cfg, "assembled", False) and not pre:  # F.py:1502-1503
            sim.model.eq_active[:] = 1
        if pre and not self._has_recipe:  # F.py:1493-1501: the listed welds are on from the start, their groups merged
            for e in pre:
                sim.model.eq_active[e] = 1
                self._merge_groups(int(m.eq_part1[e]), int(m.eq_part2[e]))
        init = getattr(self, "_init_qpos", None)
        if init is not None:
            # F.py:1505-1519, 1568-1569, 1617-1618 (set_init_qpos): the given state replaces placement, settling and the robot
            # initialisation -- no draw is taken from the RNG stream; set_env_state + forward, twice, then the common tail
            for _ in range(2):
                for i in range(self.nparts):
                    self._stop_object(i, gravity=0)
                sim.data.qpos[:] = init["qpos"]
                sim.data.qvel[:] = init["qvel"]
                sim.data.ctrl[:] = 0
                sim.model.geom_contype[robot] = ct0[robot]
                sim.model.geom_conaffinity[robot] = ca0[robot]
                sim.forward()
        else:
            # placement (init_pos is sampled on first reset and re-sampled afterwards: fix_init=False)
            if self.init_pos is None or not getattr(self.cfg, "fix_init", False):  # F.py:1518-1525
                self.init_pos, self.init_quat = sample_placement(self._rng, m, self.cfg)
            pos, quat = self.init_pos, self.init_quat
            if getattr(self.cfg, "assembled", False):  # F.py:1526-1530: one group, the parts stay at the XML's assembled poses
                self._group = [0] * self.nparts
                self.reset_draws["part_qpos"] = np.asarray(m.part_initqpos, dtype=np.float64).copy()
            else:
                self.reset_draws["part_qpos"] = np.array([np.concatenate([pos[i], quat[i]]) for i in range(self.nparts)])
                for i in range(self.nparts):
                    self._set_part_qpos(i, pos[i], quat[i])
            self._settle()
            if self._has_recipe:
                self._preassemble(pre, auto_align=True)
                self._settle()
            if self.agent != "Cursor":
                self._gravity_comp()
            self._initialize_robot_pos()
            self._fs()
            sim.model.geom_contype[robot] = ct0[robot]
            sim.model.geom_conaffinity[robot] = ca0[robot]
            if self.agent != "Cursor":
                self._gravity_comp()
            for _ in range(100):
                self._initialize_robot_pos()
                self._fs()
        sim.data.ctrl[:] = 0
        sim.data.qfrc_applied[:] = 0
        sim.data.xfrc_applied[:] = 0
        sim.data.qacc[:] = 0
        sim.data.qacc_warmstart[:] = 0
        sim.data.time[0] = 0
        sim.forward()
        if self.agent != "Cursor":
            self._gravity_comp()
        for _ in range(100):
            self._fs()
        if self._ik:
            # F.py:1643-1650: _initial_right_hand_quat = _right_hand_quat (xyzw, hand orientation in the robot base frame);
            # controller.sync_state(): the IK target position := the IK chain's own forward kinematics at the current joints
            na = len(self.arms)
            self._initial_hand_quat = [self._hand_quat(a) for a in range(na)]
            self._ik_tp = [self._IK.fk(m, sim.data.qpos[m.arm_qposadr[7 * a:7 * a + 7]], a)[0] for a in range(na)]
            self._initial_right_hand_quat, self._ik_target_pos = self._initial_hand_quat[0], self._ik_tp[0]
        self._get_next_subtask()
        # _after_reset
        self._episode_reward = 0
        self._episode_length = 0
        self._terminal = False
        self._success = False
        self._fail = False
        if self._dense is not None:
            self._dense.success_num_conn, self._dense.n_pre = self._success_num_conn, len(pre)
            self._dense.reset(self._dense_obs)  # _reset_reward_variables (furniture_sawyer_dense.py:218-220)
        return self._get_obs()

    def _preassemble(self, pre, auto_align):
        """F.py:1542-1557: recipe steps `pre` are connected during the reset -- _connect(site2, site1) with the recipe's angle"""
        m = self.m
        sites, conn = list(m.meta["site_names"]), [int(x) for x in m.conn_siteid]
        for i in pre:
            row = m.meta["site_recipe"][i]
            k1, k2 = conn.index(sites.index(row[0])), conn.index(sites.index(row[1]))
            self._target_connector_xquat = self._project_connector_quat(k2, k1, row[2] if len(row) == 3 else None)
            self._connect(k2, k1, auto_align=auto_align)
            self._connected = False
            self._connected_body1 = None

    def _settle(self):
        for _ in range(10):
            for i in range(self.nparts):
                self._stop_object(i, gravity=0)
            for _ in range(10):
                self._fs()
                for i in range(self.nparts):
                    self._slow_object(i)

    def _get_next_subtask(self):
        for k in range(self.m.neq):
            p1, p2 = self.m.eq_part1[k], self.m.eq_part2[k]
            if self._find_group(p1) != self._find_group(p2):
                self._subtask_part1, self._subtask_part2 = int(p1), int(p2)
                return
        self._subtask_part1 = self._subtask_part2 = -1

    # ---- observation (F.py:1344-1387, furniture_sawyer.py:103-155, furniture_baxter.py:98-165) ---------
    def _get_obs(self):
        m, d = self.m, self.sim.data
        ob = {}
        obj = []
        cfg = getattr(self, "cfg", None)
        ob_all = getattr(cfg, "object_ob_all", True)
        for i in range(self.nparts):
            if ob_all or i in (self._subtask_part1, self._subtask_part2):   # F.py:1363-1372
                b = m.part_bodyid[i]
                obj += [d.xpos[b].copy(), d.xquat[b].copy()]
        if not ob_all and self._subtask_part1 == -1:
            obj.append(np.zeros(14))                                         # F.py:1374-1375
        ob["object_ob"] = np.concatenate(obj)
        if getattr(cfg, "subtask_ob", False):                                # F.py:1382-1385
            ob["subtask_ob"] = np.array([self._subtask_part1 + 1, self._subtask_part2 + 1])
        if self.agent == "Cursor":
            # furniture_cursor.py:88-109: [cursor0 pos, cursor1 pos, selected0, selected1]
            ob["robot_ob"] = np.concatenate([self._cursor_pos(0), self._cursor_pos(1),
                                             np.array([s is not None for s in self._cursor_selected], dtype=float)])
            return ob
        rs = []
        nj = len(m.arm_qposadr) // len(self.arms)
        for a in range(len(self.arms)):
            site = m.eef_siteid[a]
            velp, velr = self.sim.site_vel(site)
            if getattr(self, "_ctrl", None) is None and not getattr(self, "_ik", False):  # joint_pos / joint_vel only for impedance / torque (furniture_sawyer.py:112-124)
                rs += [d.qpos[m.arm_qposadr[a * nj:(a + 1) * nj]], d.qvel[m.arm_dofadr[a * nj:(a + 1) * nj]]]
            rs += [d.qpos[m.grip_qposadr[2 * a:2 * a + 2]], d.site_xpos[site].copy(),
                   T.convert_quat(d.xquat[m.hand_bodyid[a]], to="xyzw"), velp, velr]
        ob["robot_ob"] = np.concatenate([np.asarray(x, dtype=float).ravel() for x in rs])
        return ob

    def flat_obs(self, ob):
        return np.concatenate([ob[k] for k in ("object_ob", "subtask_ob", "robot_ob") if k in ob])

    # ---- alignment / connect (F.py:847-1153) ------------------------------------------------
    def _is_aligned(self, k1, k2):
        m, cfg = self.m, self.cfg
        s1, s2 = m.conn_siteid[k1], m.conn_siteid[k2]
        p1, p2 = self.sim.data.site_xpos[s1].copy(), self.sim.data.site_xpos[s2].copy()
        R1, R2 = self.sim.data.site_xmat[s1].reshape(3, 3), self.sim.data.site_xmat[s2].reshape(3, 3)
        up1, up2, f1, f2 = R1[:, 2].copy(), R2[:, 2].copy(), R1[:, 1].copy(), R2[:, 1].copy()
        pos_dist = T.l2_dist(p1, p2)
        rot_up = T.cos_siml(up1, up2)
        with np.errstate(all="ignore"):
            proj12 = np.dot(up1, T.unit_vector(p2 - p1))
            proj21 = np.dot(up2, T.unit_vector(p1 - p2))
        angles = list(m.conn_angles[k1][: m.conn_nangle[k1]])
        if len(angles) == 0:
            fwd_ok = True
            cs = T.cos_siml(f1, f2)
            rp = T.rotate_vector_cos_siml(f1, up1, cs, 1)
            rn = T.rotate_vector_cos_siml(f1, up1, cs, -1)
            fr = rp if T.cos_siml(rp, f2) > T.cos_siml(rn, f2) else rn
            self._target_connector_xquat = T.convert_quat(T.lookat_to_quat(up1, fr), "wxyz")
        else:
            fwd_ok = False
            for ang in angles:
                fr = T.rotate_vector(f1, up1, ang)
                if T.cos_siml(fr, f2) > cfg.alignment_rot_dist_forward:
                    fwd_ok = True
                    self._target_connector_xquat = T.convert_quat(T.lookat_to_quat(up1, fr), "wxyz")
                    break
        if (pos_dist < cfg.alignment_pos_dist and rot_up > cfg.alignment_rot_dist_up and fwd_ok
                and abs(proj12) > cfg.alignment_project_dist and abs(proj21) > cfg.alignment_project_dist):
            return True
        if pos_dist < cfg.alignment_pos_dist / 2 and rot_up > cfg.alignment_rot_dist_up and fwd_ok:
            return True
        return False

    def _move_objects_translation_quat(self, part, translation, target_quat, gravity=1):
        base = self._part_qpos(part)
        g = self._find_group(part)
        for i in range(self.nparts):
            if self._find_group(i) == g:
                np_, nq = T.transform_to_target_quat(base, self._part_qpos(i), target_quat)
                self._set_part_qpos(i, np_ + translation, nq)
                self._stop_object(i, gravity=gravity)

    def _move_objects_target(self, part, target_pos, target_quat, gravity=1):
        base = self._part_qpos(part)
        self._move_objects_translation_quat(part, np.asarray(target_pos) - base[:3], target_quat, gravity)

    def _move_site_to_target(self, k_site, target_qpos, gravity=1):
        site = self.m.conn_siteid[k_site]
        qpos_base = self._site_xpos_xquat(site)
        part = self.m.conn_partid[k_site]
        body_qpos = self._part_qpos(part)
        new_pos, new_quat = T.transform_to_target_quat(qpos_base, body_qpos, target_qpos[3:])
        new_site_pos, _ = T.transform_to_target_quat(body_qpos, qpos_base, new_quat)
        self._move_objects_translation_quat(part, target_qpos[:3] - new_site_pos, new_quat, gravity)

    def _bounding_box(self, part):
        g = self._find_group(part)
        mn, mx = np.zeros(3), np.zeros(3)
        for i in range(self.nparts):
            if self._find_group(i) != g:
                continue
            a, n = self.m.part_site_adr[i], self.m.part_site_num[i]
            for s in self.m.part_sites[a:a + n]:
                p = self.sim.data.site_xpos[s]
                mn, mx = np.minimum(mn, p), np.maximum(mx, p)
        return mn, mx

    def _is_inside(self, part):
        self._fs()
        mn, mx = self._bounding_box(part)
        b = self.cfg.cursor_boundary
        return not ((mn < np.array([-b, -b, -0.05])).any() or (mx > np.array([b, b, b])).any())

    def _move_rotate_object(self, part, move_offset, rotate_offset):
        base = self._part_qpos(part)
        target = T.euler_to_quat(rotate_offset, base[3:])
        g = self._find_group(part)
        old = {}
        for i in range(self.nparts):
            if self._find_group(i) == g:
                old[i] = self._part_qpos(i)
                np_, nq = T.transform_to_target_quat(base, self._part_qpos(i), target)
                self._set_part_qpos(i, np_ + move_offset, nq)
        if self._is_inside(part):
            return True
        for i, q in old.items():
            self._set_part_qpos(i, q[:3], q[3:])
        return False

    def _connect(self, k1, k2, auto_align=True):
        m = self.m
        self._connected_sites.update([k1, k2])
        self._site1_id, self._site2_id = int(m.conn_siteid[k1]), int(m.conn_siteid[k2])
        pA, pB = int(m.conn_partid[k1]), int(m.conn_partid[k2])
        gA, gB = self._find_group(pA), self._find_group(pB)
        for g in range(m.ngeom):
            p = m.body_partid[m.geom_bodyid[g]]
            if p < 0:
                continue
            if self._find_group(int(p)) in (gA, gB) and self.sim.model.geom_contype[g] != 0:
                self.sim.model.geom_contype[g] = (1 << 30) - 1 - (1 << (gA + 1))
                self.sim.model.geom_conaffinity[g] = 1 << (gA + 1)
        if auto_align:
            tq = self._site_xpos_xquat(m.conn_siteid[k1])
            tq[3:] = self._target_connector_xquat
            self._move_site_to_target(k2, tq, self._gravity_compensation)
        if self.agent == "Cursor":
            self._stop_selected_objects()
        self._fs()
        mn1, _ = self._bounding_box(pA)
        mn2, _ = self._bounding_box(pB)
        mz = min(mn1[2], mn2[2])
        if mz < 0:
            self._move_rotate_object(pA, [0, 0, -mz], [0, 0, 0])
            self._move_rotate_object(pB, [0, 0, -mz], [0, 0, 0])
        if self.agent == "Cursor":
            self._stop_selected_objects()
        self._fs()
        # _activate_weld (F.py:2761-2776)
        for i in range(m.neq):
            p1, p2 = int(m.eq_part1[i]), int(m.eq_part2[i])
            if p1 in (pA, pB) and p2 in (pA, pB):
                self.sim.model.eq_data[i] = T.rel_pose(self._part_qpos(p1), self._part_qpos(p2))
                self.sim.model.eq_active[i] = 1
                self._merge_groups(pA, pB)
        if self.agent == "Cursor":
            self._cursor_selected[1] = None
        self._num_connected += 1
        self._connected = True
        self._connected_body1 = pA
        q = self._part_qpos(pA)
        self._connected_body1_pos, self._connected_body1_quat = q[:3], q[3:]
        self._get_next_subtask()
        if getattr(self.cfg, "reset_robot_after_attach", False):  # F.py:919-925: reset robot arm (one more draw of the env's ONE RandomState)
            self._initialize_robot_pos(record=self.attach_draws)
            if self._ik:  # controller.sync_state(): the IK target position := the chain's forward kinematics at the new joints
                self._ik_tp = [self._IK.fk(m, self.sim.data.qpos[m.arm_qposadr[7 * a:7 * a + 7]], a)[0] for a in range(len(self.arms))]
                self._ik_target_pos = self._ik_tp[0]

    def _try_connect(self, part1=None, part2=None):
        m = self.m
        g1 = None if part1 is None else self._find_group(part1)
        g2 = None if part2 is None else self._find_group(part2)
        sites1 = [k for k in range(len(m.conn_siteid)) if g1 is None or self._find_group(int(m.conn_partid[k])) == g1]
        sites2 = [k for k in range(len(m.conn_siteid)) if g2 is None or self._find_group(int(m.conn_partid[k])) == g2]
        if not sites1 or not sites2:
            return False
        ids1 = set(range(self.nparts)) if g1 is None else {i for i in range(self.nparts) if self._find_group(i) == g1}
        ids2 = set(range(self.nparts)) if g2 is None else {i for i in range(self.nparts) if self._find_group(i) == g2}
        both = ids1 | ids2
        if not any(int(m.eq_part1[i]) in both and int(m.eq_part2[i]) in both for i in range(m.neq)):
            return False
        for k1 in sites1:
            for k2 in sites2:
                if k1 in self._connected_sites or k2 in self._connected_sites:
                    continue
                a1, b1, a2, b2 = m.conn_keya[k1], m.conn_keyb[k1], m.conn_keya[k2], m.conn_keyb[k2]
                match = (b1 < 0 and b2 < 0 and a1 == a2) if (b1 < 0 or b2 < 0) else (a1 == b2 and b1 == a2)
                if not match:
                    continue
                if self._is_aligned(k1, k2):
                    if self._connect_step < self._num_connect_steps:
                        s1 = m.conn_siteid[k1]
                        target_pos = self._site_xpos_xquat(s1)[:3]
                        site1_quat = self._target_connector_xquat
                        p2 = int(m.conn_partid[k2])
                        p2q = self._part_qpos(p2)
                        s2pq = self._site_xpos_xquat(m.conn_siteid[k2])
                        body_pos, body_rot = T.transform_to_target_quat(s2pq, p2q, site1_quat)
                        body_pos = body_pos + (target_pos - s2pq[:3])
                        if self._connect_step == 0:
                            n = self._num_connect_steps
                            self.next_rot = [T.quat_slerp(p2q[3:], body_rot, (f + 1) / n) for f in range(n)]
                            xs = np.linspace(1 / n, 0.9, n)
                            self.next_pos = [p2q[:3] + x * (body_pos - p2q[:3]) for x in xs]
                        self._move_objects_target(p2, self.next_pos[self._connect_step], list(self.next_rot[self._connect_step]))
                        self._connect_step += 1
                        return False
                    self._connect(k1, k2, self.cfg.auto_align)
                    self._connect_step = 0
                    self.next_pos = self.next_rot = None
                    return True
        self._connect_step = 0
        return False

    # ---- step (F.py:364-449) ----------------------------------------------------------------------
    def _touch_sets(self):
        """per arm: (left-finger-touch, right-finger-touch, floor-touch) part sets from data.contact."""
        m = self.m
        out = []
        for a in range(len(self.arms)):
            L, R, Fl = set(), set(), set()
            for g1, g2 in self.sim.contacts():
                for ga, gb in ((g1, g2), (g2, g1)):
                    p = m.body_partid[m.geom_bodyid[gb]]
                    if p < 0:
                        continue
                    role = m.geom_fingerrole[ga]
                    if role & (1 << (2 * a)):
                        L.add(int(p))
                    if role & (1 << (2 * a + 1)):
                        R.add(int(p))
                    if ga == m.floor_geomid[0]:
                        Fl.add(int(p))
            out.append((L, R, Fl))
        return out

    def _connect_scan(self):
        """F.py:1290-1330: per arm, the first part (in part order) touched by both fingers is tried; a successful connect ends
        the scan, a failed one only ends this arm's (quirk Q4)."""
        for (L, R, _) in self._touch_sets():
            for i in range(self.nparts):
                if i in L and i in R:
                    if self._try_connect(i):
                        return
                    break

    def _setup_action(self, action):
        m = self.m
        if self.cfg.rescale_actions:
            action = np.clip(action, -1, 1)
        na = len(m.arm_qposadr)
        arm = action[:na]
        grips = []
        for a in range(len(self.arms)):
            g = action[na + a]
            grips += [g, -g]
        act = np.concatenate([arm, grips])
        if self.cfg.rescale_actions:
            act = m.ctrl_bias + m.ctrl_weight * act
        self._gravity_comp()
        return act

    def _do_simulation(self, ctrl):
        try:
            if self.m.nu:
                self.sim.data.ctrl[:] = 0 if ctrl is None else ctrl
            if self.agent == "Cursor":
                sel = [self._find_group(s) for s in self._cursor_selected if s is not None]
                for i in range(self.nparts):
                    self._stop_object(i, gravity=1 if self._find_group(i) in sel else 0)
            self.sim.forward()
            for _ in range(self._n_substeps):
                self.sim.step()
            if self.agent == "Cursor":
                for i in range(self.nparts):
                    if self._find_group(i) in sel:
                        self._stop_object(i, gravity=1)
        except SimUnstable:
            self.reset()
            self._fail = True

    def _hand_quat(self, arm=0):
        """F.py:3380-3457: mat2quat of the <arm>_hand orientation in the frame of the body 'base' (data of the last forward pass)."""
        m, d = self.m, self.sim.data
        Rb = self._IK.q2m(np.asarray(m.ik_base_quat, float))
        return T.mat2quat(Rb.T @ d.xmat[int(m.hand_bodyid[arm])].reshape(3, 3))

    def _right_hand_quat(self):
        return self._hand_quat(0)

    def _do_ik_step(self, action):
        """F.py:2899-3063 (control_type 'ik' / 'ik_quaternion', Sawyer and Baxter) over oracle/ik.py instead of pybullet.  Note the
        reference feeds the xyzw quaternion `_initial_<arm>_hand_quat` to euler_to_quat, whose pyquaternion reads it as wxyz
        (F.py:2917-2919): the same functions are used here in the same way, so the commanded orientation is garbled identically."""
        IK, m, d = self._IK, self.m, self.sim.data
        sens, gain, rest_current, rz = [float(v) for v in m.ik_params]
        na = len(self.arms)
        grips, qcmd = [], []
        for a in range(na):
            d_pos, rotation, self._initial_hand_quat[a], g = IK.preprocess(
                self.cfg.control_type, action, self.cfg.move_speed, self.cfg.rotate_speed, d.xpos[int(m.hand_bodyid[a])],
                self._hand_quat(a), self._initial_hand_quat[a], self.agent, a)
            grips.append(g)
            # <Robot>IKController.get_control -> joint_positions_for_eef_command (sawyer_ik_controller.py:51-88, 227-269;
            # baxter_ik_controller.py:47-100, 296-333): the target moves by dpos * user_sensitivity in the base frame
            self._ik_tp[a] = self._ik_tp[a] + d_pos * sens
            target_R = rotation @ IK.rot_z(-np.pi / 2) if rz else rotation
            qa = d.qpos[m.arm_qposadr[7 * a:7 * a + 7].astype(int)]
            qcmd.append(IK.solve(m, qa, self._ik_tp[a], target_R, arm=a, rest=qa.copy() if rest_current else None))
        self._ik_q_cmd = np.concatenate(qcmd)
        self._initial_right_hand_quat, self._ik_target_pos = self._initial_hand_quat[0], self._ik_tp[0]
        arm_q = m.arm_qposadr.astype(int)
        for i in range(self._action_repeat):
            vel = IK.velocities(d.qpos[arm_q], self._ik_q_cmd, gain)
            ctrl = self._setup_action(np.concatenate([vel, grips]))
            self._do_simulation(ctrl)
            if self._fail:
                break

    def _do_controller_step(self, action):
        """F.py:3065-3093 + _pre_action (F.py:1706-1759).  update_model (arm_controller.py:109-136) reads MuJoCo's memory as
        sim.step() left it: poses / Jacobian / mass matrix / qfrc_bias of the forward pass BEFORE the last integration, qpos and
        qvel after it; body_xvelp/xvelr are mujoco_py's jac . qvel."""
        C, m, sim = self._C, self.m, self.sim
        act = C.preprocess_action(action, self.cfg.move_speed)
        arm_q, arm_d, grip_d = m.arm_qposadr.astype(int), m.arm_dofadr.astype(int), m.grip_dofadr.astype(int)
        try:
            sim.forward()
            for i in range(self._n_substeps):
                d = sim.data
                hb = int(m.hand_bodyid[0])
                jp, jr = sim.body_jac(hb, d.xpos[hb])
                model = dict(pos=d.xpos[hb].copy(), mat=d.xmat[hb].reshape(3, 3).copy(), velp=jp @ d.qvel, velr=jr @ d.qvel,
                             q=d.qpos[arm_q].copy(), qd=d.qvel[arm_d].copy(), Jx=jp[:, arm_d], Jr=jr[:, arm_d],
                             M=sim.full_M()[np.ix_(arm_d, arm_d)])
                arm, grip = C.pre_action_ctrl(self._ctrl[0], act, i == 0, model, d.qfrc_bias[arm_d], m.ctrl_bias[grip_d],
                                              m.ctrl_weight[grip_d], float(m.opt[0]))
                d.ctrl[arm_d] = arm      # (the reference indexes ctrl with the joint-velocity indices, F.py:1756)
                d.ctrl[grip_d] = grip
                sim.step()
        except SimUnstable:
            self.reset()
            self._fail = True

    def step(self, action):
        action = np.asarray(action, dtype=np.float64).copy()
        self._connected = False
        a = action.copy()
        if self.agent == "Sawyer" and self.cfg.discrete_grip:
            a[-2] = -1 if action[-2] < 0 else 1
        if self.agent == "Cursor":
            self._step_discrete(a.copy())
            self._do_simulation(None)
        else:
            connect = a[-1]
            if self._ik:
                self._do_ik_step(a)
            elif self._ctrl is not None:
                self._do_controller_step(a)
            else:
                ctrl = self._setup_action(a[:-1])
                self._do_simulation(ctrl)
            if connect > 0:
                self._connect_scan()
        if self._connected_body1 is not None:
            self.sim.forward()
            self._move_objects_target(self._connected_body1, self._connected_body1_pos, self._connected_body1_quat, self._gravity_compensation)
            self._connected_body1 = None
            self._fs()
        ob = self._get_obs()
        done = False
        if self._num_connected == self._success_num_conn and self.nparts > 1:
            self._success = True
            done = True
        if self._dense is not None:
            # FurnitureSawyerEnv._step: done = done or _done of the dense _compute_reward, which also owns _success (:78-79)
            reward, d2, self._success, info = self._dense.compute(
                action, lambda st: self._is_aligned(self._dsub[st]["k_leg"], self._dsub[st]["k_table"]), self._connected)
            done = done or d2
        else:
            reward, info = self._compute_reward(action)
        fail = self._fail
        done, penalty = self._after_step(reward, done)
        info.update(num_connected=self._num_connected, success=int(self._success), fail=int(fail), site1=self._site1_id,
                    site2=self._site2_id, episode_length=self._episode_length, connected_this_step=int(self._connected))
        return ob, reward + penalty, done, info

    def _after_step(self, reward, done):
        """F.py:451-480: episode counters, time limit by EQUALITY (quirk Q9), failure -> terminal with a one-shot penalty.
        -> (terminal, penalty); step_log's episode_reward is _episode_reward + penalty."""
        self._episode_reward += reward
        self._episode_length += 1
        penalty = 0
        if self._episode_length == self.cfg.max_episode_steps or self._fail:
            done = True
            if self._fail:
                self._fail = False
                penalty = -self.cfg.unstable_penalty_coef
        return done, penalty

    def _compute_reward(self, ac):
        touch = pick = 0.0
        if self.agent != "Cursor":
            for (L, R, Fl) in self._touch_sets():
                for i in range(self.nparts):
                    if i in L and i in R:
                        if not self._touched[i]:
                            self._touched[i] = True
                            touch += self.cfg.touch_reward
                        if i not in Fl and not self._picked[i]:
                            self._picked[i] = True
                            pick += self.cfg.pick_reward
        succ = self.cfg.success_reward * (self._num_connected - self._prev_num_connected)
        self._prev_num_connected = self._num_connected
        ctrl = 0.0 if self.agent == "Cursor" else -self.cfg.ctrl_penalty_coef * float(np.square(ac).sum())
        return succ + touch + pick + ctrl, dict(success_reward=succ, touch_reward=touch, pick_reward=pick, ctrl_penalty=ctrl)

    # ---- Cursor agent (furniture_cursor.py, F.py:700-845) --------------------------------------------
    def _cursor_pos(self, k):
        return self.sim.data.xpos[self.m.cursor_bodyid[k]].copy()

    def _move_cursor(self, k, off):
        pos = self._cursor_pos(k) + off
        b = self.cfg.cursor_boundary
        if (np.abs(pos) < b).all() and pos[2] >= self.cfg.move_speed * 0.45:
            self.sim.model.body_pos[self.m.cursor_bodyid[k]] = pos
            return True
        return False

    def _stop_selected_objects(self, gravity=1):
        sel = [self._find_group(s) for s in self._cursor_selected if s is not None]
        for i in range(self.nparts):
            if self._find_group(i) in sel:
                self._stop_object(i, gravity)

    def _on_collision(self, k, part):
        """on_collision('cursorK', part name): substring match on geom names (F.py:3290-3310)."""
        names = self.m.meta["geom_names"]
        ref, body = "cursor%d" % k, self.m.meta["part_names"][part]
        for g1, g2 in self.sim.contacts():
            n1, n2 = names[g1], names[g2]
            if (ref in n1 or ref in n2) and (body in n1 or body in n2):
                return True
        return False

    def _select_object(self, k):
        for i in range(self.nparts):
            g = self._find_group(i)
            if any(s is not None and g == self._find_group(s) for s in self._cursor_selected):
                continue
            if self._on_collision(k, i):
                return i
        return None

    def _step_discrete(self, a):
        """furniture.py:800-845."""
        assert len(a) == 15
        actions = [a[:7], a[7:]]
        for k in range(2):
            move = actions[k][0:3] * self.cfg.move_speed
            rot = actions[k][3:6] * self.cfg.rotate_speed
            select = actions[k][6] > 0
            if not select:
                self._cursor_selected[k] = None
            if not self._move_cursor(k, move):
                continue
            if self._cursor_selected[k] is not None:
                if not self._move_rotate_object(self._cursor_selected[k], move, rot):
                    self._move_cursor(k, -move)
                    continue
            if select and self._cursor_selected[k] is None:
                self._cursor_selected[k] = self._select_object(k)
        connect = a[14]
        # the reference stores part NAMES here, so its truthiness test is an `is not None` test
        if connect > 0 and self._cursor_selected[0] is not None and self._cursor_selected[1] is not None:
            self._try_connect(self._cursor_selected[0], self._cursor_selected[1])
        elif self._connect_step > 0:
            self._connect_step = 0

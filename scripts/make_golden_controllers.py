"""Generate tests/golden/controllers.npz from the REFERENCE's own arm controllers.

Runs only in the build container (needs /root/reference).  `furniture/env/controllers/arm_controller.py` is plain numpy /
scipy code once `mujoco_py` is stubbed: every quantity it reads from the simulator (`update_model`, :109-136) is injected
through a fake `sim`, and `mujoco_py.cymj._mj_fullM` is replaced by a copy of a dense matrix we supply.  The five
controller classes are constructed exactly as `FurnitureEnv._load_controller` does (`furniture.py:1665-1704`): parameters
from `controllers/controller_config.hjson`, no overrides.  Recorded per physics substep: the injected inputs and the torques
returned by `action_to_torques(action, policy_step)`.

A second block pins `FurnitureEnv._do_controller_step` / `_pre_action` (`furniture.py:1706-1759, 3065-3093`) themselves: the
reference methods run on a fake `self` whose `sim.step()` only advances a scripted state; the `ctrl` vector they write at
every substep is recorded.
"""
import json
import os
import re
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden_env_logic import import_reference, rand_rot  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "controllers.npz")
TYPES = ["position_orientation", "position", "joint_impedance", "joint_velocity", "joint_torque"]
NV, NARM = 12, 7
HAND = "right_hand"


def load_hjson(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    return json.loads(txt)


class FakeModel:
    def __init__(self):
        self.opt = types.SimpleNamespace(timestep=0.002)
        self.actuator_ctrlrange = np.array([[-80.0, 80.0]] * 5 + [[-12.0, 12.0]] * 2 + [[-0.0115, 0.020833], [-0.020833, 0.0115]])

    def body_name2id(self, name):
        assert name == HAND
        return 3


class FakeData:
    def __init__(self):
        self.body_xpos = np.zeros((5, 3))
        self.body_xmat = np.zeros((5, 9))
        self.body_xvelp = np.zeros((5, 3))
        self.body_xvelr = np.zeros((5, 3))
        self.qpos = np.zeros(NV + 3)
        self.qvel = np.zeros(NV)
        self.qM = np.zeros(NV * NV)
        self.jacp = np.zeros((3, NV))
        self.jacr = np.zeros((3, NV))
        self.ctrl = np.zeros(9)
        self.qfrc_bias = np.zeros(NV)

    def get_body_jacp(self, name):
        assert name == HAND
        return self.jacp.reshape(-1).copy()

    def get_body_jacr(self, name):
        assert name == HAND
        return self.jacr.reshape(-1).copy()


class FakeSim:
    def __init__(self):
        self.model, self.data = FakeModel(), FakeData()
        self.calls = []

    def forward(self):
        self.calls.append("forward")

    def step(self):
        self.calls.append("step")
        if self.on_step is not None:
            self.on_step()

    on_step = None


def spd(rng, n, lo=0.3, hi=3.0):
    q, _ = np.linalg.qr(rng.randn(n, n))
    return (q * rng.uniform(lo, hi, n)) @ q.T


class Scene:
    """A smooth scripted 'robot': joint state random walk, hand pose drifting, Jacobian and mass matrix slowly varying."""

    def __init__(self, rng, singular=False, fast=False):
        self.rng = rng
        self.q = rng.uniform(-1, 1, NARM)
        self.qd = rng.uniform(-0.5, 0.5, NARM) * (12.0 if fast else 1.0)
        self.pos = rng.uniform(-0.5, 0.5, 3) + np.array([0.5, 0, 1.0])
        self.R = rand_rot(rng)
        self.J = rng.uniform(-0.6, 0.6, (6, NV))
        self.J[:, NARM:] = 0
        if singular:
            # Jx and Jr nearly lose a direction: one singular value of Jx M^-1 Jx' (Jr ...) falls below 0.00025 and is zeroed
            # (:782-790); exactly singular would make the reference's own scipy.linalg.inv of the 6x6 raise (:770)
            self.J[2, :NARM] *= 2e-3
            self.J[3, :NARM] *= 1e-2
        self.M = spd(rng, NV)

    def advance(self):
        rng = self.rng
        self.qd = self.qd + rng.uniform(-0.02, 0.02, NARM)
        self.q = self.q + 0.002 * self.qd
        self.pos = self.pos + rng.uniform(-2e-4, 2e-4, 3)
        w = rng.uniform(-2e-3, 2e-3, 3)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        u, _, vt = np.linalg.svd((np.eye(3) + K) @ self.R)
        self.R = u @ vt
        self.J = self.J * (1 + rng.uniform(-1e-3, 1e-3, self.J.shape))
        d = rng.uniform(-1e-3, 1e-3, (NV, NV))
        self.M = self.M + d + d.T

    def write(self, sim):
        d = sim.data
        d.body_xpos[3] = self.pos
        d.body_xmat[3] = self.R.reshape(-1)
        qv = np.zeros(NV)
        qv[:NARM] = self.qd
        d.body_xvelp[3] = self.J[:3] @ qv
        d.body_xvelr[3] = self.J[3:] @ qv
        d.qpos[:NARM] = self.q
        d.qvel[:NARM] = self.qd
        d.jacp[:] = self.J[:3]
        d.jacr[:] = self.J[3:]
        d.qM[:] = self.M.reshape(-1)

    def record(self):
        qv = np.zeros(NV)
        qv[:NARM] = self.qd
        return np.concatenate([self.pos, self.R.reshape(-1), self.J[:3] @ qv, self.J[3:] @ qv, self.q, self.qd,
                               self.J[:3, :NARM].reshape(-1), self.J[3:, :NARM].reshape(-1), self.M[:NARM, :NARM].reshape(-1)])


def main():
    F = import_reference()
    import mujoco_py
    import furniture.env.controllers.arm_controller as AC

    def full_m(model, dst, qM):
        dst[:] = np.asarray(qM).reshape(-1)

    mujoco_py.cymj = types.SimpleNamespace(_mj_fullM=full_m)
    AC.mujoco_py = mujoco_py
    params = load_hjson("/root/reference/furniture/env/controllers/controller_config.hjson")
    ctor = {"position_orientation": AC.PositionOrientationController, "position": AC.PositionController,
            "joint_impedance": AC.JointImpedanceController, "joint_velocity": AC.JointVelocityController,
            "joint_torque": AC.JointTorqueController}
    joint_index = list(range(NARM))
    out = {"types": np.array(TYPES), "params_json": np.array(json.dumps({k: params[k] for k in TYPES}))}
    rng = np.random.RandomState(20260924)
    n_policy, n_sub = 3, 10
    for ti, tname in enumerate(TYPES):
        for si in range(3):
            scene = Scene(rng, singular=(si == 1), fast=(si == 2))
            sim = FakeSim()
            c = ctor[tname](**dict(params[tname]))
            c.reset()
            cd = c.control_dim
            ins, acts, flags, outs = [], [], [], []
            for p in range(n_policy):
                a = rng.uniform(-1.3, 1.3, cd)  # beyond [-1, 1]: transform_action clips (:99-107)
                for s in range(n_sub):
                    scene.write(sim)
                    c.update_model(sim, id_name=HAND, joint_index=joint_index)
                    tq = np.array(c.action_to_torques(a.copy(), s == 0), dtype=np.float64)
                    ins.append(scene.record())
                    acts.append(a)
                    flags.append(s == 0)
                    outs.append(tq)
                    scene.advance()
            k = "%s_%d" % (tname, si)
            out[k + "_in"], out[k + "_act"] = np.array(ins), np.array(acts)
            out[k + "_policy"], out[k + "_torque"] = np.array(flags), np.array(outs)
            out[k + "_interp_steps"] = np.array(c.interpolation_steps)

    # ---- FurnitureEnv._do_controller_step / _pre_action on a fake self -------------------------------------------------
    TG = __import__("furniture.env.models.grippers.two_finger_gripper", fromlist=["TwoFingerGripper"])
    for tname in ("position_orientation", "joint_velocity"):
        scene = Scene(rng)
        sim = FakeSim()
        c = ctor[tname](**dict(params[tname]))
        c.reset()
        env = types.SimpleNamespace()
        env._control_type = tname
        env._agent_type = "Sawyer"
        env._arms = ["right"]
        env._move_speed = 0.1
        env._control_timestep, env._model_timestep, env._cur_time = 0.02, 0.002, 0.0  # 10 substeps keep the fixture small
        env.sim = sim
        env.controller = {"right": c}
        env.gripper = {"right": types.SimpleNamespace(dof=1, format_action=lambda a: TG.TwoFingerGripper.format_action(None, a))}
        env._ref_joint_pos_indexes = {"right": joint_index}
        env._ref_joint_vel_indexes = {"right": joint_index}
        env._ref_gripper_joint_vel_indexes = {"right": [7, 8]}
        env._pre_action = types.MethodType(F.FurnitureEnv._pre_action, env)
        ctrls, biases, ins = [], [], []

        def on_step():
            ctrls.append(sim.data.ctrl.copy())
            scene.advance()
            scene.write(sim)
            sim.data.qfrc_bias[:] = rng.uniform(-3, 3, NV)
            biases.append(sim.data.qfrc_bias.copy())
            ins.append(scene.record())

        sim.on_step = on_step
        scene.write(sim)
        sim.data.qfrc_bias[:] = rng.uniform(-3, 3, NV)
        bias0, in0 = sim.data.qfrc_bias.copy(), scene.record()
        acts = []
        for p in range(3):
            a = rng.uniform(-1, 1, c.control_dim + 2)
            a[c.control_dim] = -1.0 if a[c.control_dim] < 0 else 1.0  # grip already discretised by FurnitureSawyerEnv._step
            acts.append(a.copy())
            F.FurnitureEnv._do_controller_step(env, a)
        k = "envstep_%s" % tname
        out[k + "_act"], out[k + "_ctrl"] = np.array(acts), np.array(ctrls)
        out[k + "_bias"], out[k + "_in"] = np.array([bias0] + biases), np.array([in0] + ins)
        out[k + "_calls"] = np.array(sim.calls)
        out[k + "_ctrlrange"] = sim.model.actuator_ctrlrange
    # ---- FurnitureEnv._do_ik_step (furniture.py:2899-2991 / 2994-3063) on a fake self: what reaches the IK controller --------------
    # pyquaternion is absent here: the reference's transform_utils gets furniture_amd.transform_utils.Quaternion (a restatement of
    # pyquaternion's constructor / product / iteration semantics) under the name it imports, so that euler_to_quat is the
    # reference's own code.  Pinned: action scaling + permutation, _bounded_d_pos, the accumulation of _initial_right_hand_quat
    # (an xyzw quaternion handed to pyquaternion, which reads wxyz), d_quat, _make_input's rotation, the three closed-loop repeats.
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from furniture_amd.transform_utils import Quaternion as OurQuaternion
    import furniture.env.transform_utils as RT
    RT.Quaternion = OurQuaternion
    for ctype in ("ik", "ik_quaternion"):
        rec = dict(act=[], hand_pos=[], rhq=[], init_in=[], dpos=[], rot=[], init_out=[], low=[], nsim=[])
        for t in range(24):
            env = types.SimpleNamespace()
            env._control_type, env._agent_type, env._arms = ctype, "Sawyer", ["right"]
            env._move_speed, env._rotate_speed, env._action_repeat, env._record_demo = 0.1, 22.5, 3, False
            env._min_gripper_pos, env._max_gripper_pos = np.array([-1.5, -1.5, 0.0]), np.array([1.5, 1.5, 1.5])
            hand_pos = rng.uniform(-0.5, 0.5, 3) + np.array([0.0, 0.2, 0.6 if t % 4 else 0.03])  # near the z = 0 bound at times
            rhq = RT.mat2quat(rand_rot(rng).astype(np.float32))
            init_q = RT.mat2quat(rand_rot(rng).astype(np.float32))
            env._right_hand_quat, env._initial_right_hand_quat = rhq, init_q.copy()
            env.sim = types.SimpleNamespace(data=types.SimpleNamespace(get_body_xpos=lambda name, hp=hand_pos: hp))
            env._bounded_d_pos = types.MethodType(F.FurnitureEnv._bounded_d_pos, env)
            env._make_input = types.MethodType(F.FurnitureEnv._make_input, env)
            calls = dict(ctrl=[], low=[], nsim=0)
            vel_seq = [rng.uniform(-1, 1, 7) for _ in range(3)]

            def get_control(dpos=None, rotation=None, calls=calls, vel_seq=vel_seq):
                if dpos is not None:
                    calls["dpos"], calls["rot"] = np.array(dpos, dtype=float), np.array(rotation, dtype=float)
                return vel_seq[len(calls["ctrl"])]

            def setup_action(a, calls=calls):
                calls["low"].append(np.array(a, dtype=float))
                calls["ctrl"].append(1)
                return a

            def do_sim(ctrl, calls=calls):
                calls["nsim"] += 1

            env._controller = types.SimpleNamespace(get_control=get_control)
            env._setup_action, env._do_simulation = setup_action, do_sim
            a = rng.uniform(-1, 1, 8 if ctype == "ik" else 9)
            a[-2] = -1.0 if a[-2] < 0 else 1.0
            F.FurnitureEnv._do_ik_step(env, a.copy())
            rec["act"].append(a); rec["hand_pos"].append(hand_pos); rec["rhq"].append(rhq); rec["init_in"].append(init_q)
            rec["dpos"].append(calls["dpos"]); rec["rot"].append(calls["rot"]); rec["init_out"].append(np.array(env._initial_right_hand_quat, dtype=float))
            rec["low"].append(np.array(calls["low"])); rec["nsim"].append(calls["nsim"])
            rec.setdefault("vel", []).append(np.array(vel_seq))
        for k, v in rec.items():
            out["ikstep_%s_%s" % (ctype, k)] = np.array(v)
    # ---- the same for Baxter (two arms: furniture.py:2925-2958, 3000-3018; get_control(right, left)) -----------------------------------
    for ctype in ("ik", "ik_quaternion"):
        rec = dict(act=[], hand_pos=[], rhq=[], init_in=[], dpos=[], rot=[], init_out=[], low=[], nsim=[], vel=[])
        for t in range(16):
            env = types.SimpleNamespace()
            env._control_type, env._agent_type, env._arms = ctype, "Baxter", ["right", "left"]
            env._move_speed, env._rotate_speed, env._action_repeat, env._record_demo = 0.1, 22.5, 3, False
            env._min_gripper_pos, env._max_gripper_pos = np.array([-1.5, -1.5, 0.0]), np.array([1.5, 1.5, 1.5])
            hp = {"right_hand": rng.uniform(-0.5, 0.5, 3) + np.array([0.0, 0.2, 0.6]), "left_hand": rng.uniform(-0.5, 0.5, 3) + np.array([0.0, 0.2, 0.02])}
            rhq = [RT.mat2quat(rand_rot(rng).astype(np.float32)) for _ in range(2)]
            init_q = [RT.mat2quat(rand_rot(rng).astype(np.float32)) for _ in range(2)]
            env._right_hand_quat, env._left_hand_quat = rhq
            env._initial_right_hand_quat, env._initial_left_hand_quat = init_q[0].copy(), init_q[1].copy()
            env.sim = types.SimpleNamespace(data=types.SimpleNamespace(get_body_xpos=lambda name, hp=hp: hp[name]))
            env._bounded_d_pos = types.MethodType(F.FurnitureEnv._bounded_d_pos, env)
            env._make_input = types.MethodType(F.FurnitureEnv._make_input, env)
            calls = dict(ctrl=[], low=[], nsim=0)
            vel_seq = [rng.uniform(-1, 1, 14) for _ in range(3)]

            def get_control(right=None, left=None, calls=calls, vel_seq=vel_seq):
                if right is not None:
                    calls["dpos"] = np.array([right["dpos"], left["dpos"]], dtype=float)
                    calls["rot"] = np.array([right["rotation"], left["rotation"]], dtype=float)
                return vel_seq[len(calls["ctrl"])]

            def setup_action(a, calls=calls):
                calls["low"].append(np.array(a, dtype=float))
                calls["ctrl"].append(1)
                return a

            env._controller = types.SimpleNamespace(get_control=get_control)
            env._setup_action = setup_action
            env._do_simulation = lambda ctrl, calls=calls: calls.__setitem__("nsim", calls["nsim"] + 1)
            a = rng.uniform(-1, 1, 15 if ctype == "ik" else 17)
            F.FurnitureEnv._do_ik_step(env, a.copy())
            rec["act"].append(a); rec["hand_pos"].append([hp["right_hand"], hp["left_hand"]]); rec["rhq"].append(rhq); rec["init_in"].append(init_q)
            rec["dpos"].append(calls["dpos"]); rec["rot"].append(calls["rot"])
            rec["init_out"].append([np.array(env._initial_right_hand_quat, dtype=float), np.array(env._initial_left_hand_quat, dtype=float)])
            rec["low"].append(np.array(calls["low"])); rec["nsim"].append(calls["nsim"]); rec["vel"].append(np.array(vel_seq))
        for k, v in rec.items():
            out["ikstep_baxter_%s_%s" % (ctype, k)] = np.array(v)

    # ---- SawyerIKController (controllers/sawyer_ik_controller.py) around a fake pybullet: everything but the IK solve itself ---------
    # The fake `p` keeps the joint states the controller writes (resetJointState), answers getLinkState(robot, 6) with the
    # centre-of-mass frame of right_l6 from the compiled model's URDF chain (oracle/ik.fk, itself checked against the MJCF
    # kinematics) on a base at (0, 0, 0.9) as loadURDF places it, records every calculateInverseKinematics call and answers it with a
    # joint vector we choose.  Pinned: sync_state's initial target, target += dpos * user_sensitivity, the Rz(-90 deg) orientation
    # convention, the base -> world target conversion, 20 solver calls with the rest-pose / limit arguments, the P controller.
    _m2q = RT.mat2quat  # np.array(rmat, dtype=float32, copy=False) raises under numpy >= 2 for float64 input; numpy 1.x (which the
    RT.mat2quat = lambda rmat, precise=False: _m2q(np.asarray(rmat, dtype=np.float32), precise)  # reference ran on) silently copies
    import furniture.env.controllers.sawyer_ik_controller as SIK
    from furniture_amd.mjcf.model import load_compiled
    from oracle import ik as OIK
    cm = load_compiled("Sawyer", "table_lack_0825")

    class FakeBullet:
        DIRECT, POSITION_CONTROL = 0, 1

        def __init__(self):
            self.q = np.zeros(7)
            self.calls = []
            self.answer = None

        def connect(self, *a, **k): return 0
        def resetSimulation(self, *a, **k): pass
        def loadURDF(self, path, base, useFixedBase=1):
            self.base = np.array(base, dtype=float)
            self.urdf = path
            return 7
        def setRealTimeSimulation(self, *a, **k): pass
        def resetJointState(self, robot, i, v, *a): self.q[i] = v
        def getBasePositionAndOrientation(self, robot): return (tuple(self.base), (0.0, 0.0, 0.0, 1.0))
        def getLinkState(self, robot, link):
            assert link == 6
            pos, R, _, _ = OIK.fk(cm, self.q)
            return (tuple(pos + self.base), tuple(RT.mat2quat(R.astype(np.float32))))
        def calculateInverseKinematics(self, robot, link, pos, **kw):
            self.calls.append(dict(link=link, pos=np.array(pos, dtype=float), orn=np.array(kw["targetOrientation"], dtype=float),
                                   rest=np.array(kw["restPoses"], dtype=float), lower=np.array(kw.get("lowerLimits", [])),
                                   upper=np.array(kw.get("upperLimits", [])), damping=np.array(kw["jointDamping"])))
            return list(self.answer)

    fb = FakeBullet()
    SIK.p = fb
    jpos = [cm.arm_initqpos.copy()]
    ctl = SIK.SawyerIKController(bullet_data_path="/x", robot_jpos_getter=lambda: jpos[0])
    out["sik_urdf"] = np.array(fb.urdf)
    out["sik_sync_target"] = np.array(ctl.ik_robot_target_pos, dtype=float)
    out["sik_base"] = fb.base
    recs = dict(q=[], dpos=[], rot=[], answer=[], target_after=[], call_pos=[], call_orn=[], ncalls=[], vel=[], vel2=[], q2=[])
    for t in range(12):
        jpos[0] = cm.arm_initqpos + rng.uniform(-0.3, 0.3, 7)
        dpos, rot = rng.uniform(-0.1, 0.1, 3), rand_rot(rng).astype(np.float32)
        fb.answer = jpos[0] + rng.uniform(-0.4, 0.4, 7)
        fb.calls = []
        vel = ctl.get_control(dpos=dpos.copy(), rotation=rot.copy())
        jpos.append(None)
        q2 = jpos[0] + rng.uniform(-0.05, 0.05, 7)
        q_first = jpos[0].copy()
        jpos[0] = q2
        vel2 = ctl.get_control()   # closed-loop repeat: no new target
        recs["q"].append(q_first); recs["dpos"].append(dpos); recs["rot"].append(rot.astype(float)); recs["answer"].append(fb.answer.copy())
        recs["target_after"].append(np.array(ctl.ik_robot_target_pos, dtype=float)); recs["call_pos"].append(fb.calls[0]["pos"]); recs["call_orn"].append(fb.calls[0]["orn"])
        recs["ncalls"].append(len(fb.calls)); recs["vel"].append(np.array(vel)); recs["vel2"].append(np.array(vel2)); recs["q2"].append(q2)
        if t == 0:
            out["sik_rest"], out["sik_lower"], out["sik_upper"], out["sik_damping"] = fb.calls[0]["rest"], fb.calls[0]["lower"], fb.calls[0]["upper"], fb.calls[0]["damping"]
            out["sik_link"] = np.array(fb.calls[0]["link"])
        jpos = [jpos[0]]
    for k, v in recs.items():
        out["sik_" + k] = np.array(v)
    out["sik_user_sensitivity"] = np.array(ctl.user_sensitivity)

    # ---- BaxterIKController on a fake pybullet (48 joints, 15 movable: head_pan + 2 x 7) ------------------------------------------------
    import furniture.env.controllers.baxter_ik_controller as BIK
    import xml.etree.ElementTree as ET
    from furniture_amd.mjcf.urdf_chain import pybullet_joint_order
    cb = load_compiled("Baxter", "desk_mikael_1064")
    urdf_order = pybullet_joint_order(ET.parse("/root/reference/furniture/env/models/assets/bullet_data/baxter_description/urdf/baxter_mod.urdf").getroot())

    class FakeBaxterBullet:
        DIRECT, POSITION_CONTROL = 0, 1

        def __init__(self):
            self.state = np.zeros(len(urdf_order))
            self.calls, self.answers = [], None

        def connect(self, *a, **k): return 0
        def resetSimulation(self, *a, **k): pass
        def loadURDF(self, path, base, useFixedBase=1):
            self.base, self.urdf = np.array(base, dtype=float), path
            return 3
        def setRealTimeSimulation(self, *a, **k): pass
        def getNumJoints(self, robot): return len(urdf_order)
        def getJointInfo(self, robot, i):
            j = urdf_order[i]
            mov = j.get("type") != "fixed"
            lim = j.find("limit")
            lo, hi = (float(lim.get("lower")), float(lim.get("upper"))) if (mov and lim is not None) else (0.0, -1.0)
            return (i, j.get("name").encode(), 0 if mov else 4, (7 + i) if mov else -1, (6 + i) if mov else -1, 0, 0.0, 0.0, lo, hi, 0.0, 0.0)
        def getJointState(self, robot, i): return (self.state[i], 0.0, (0,) * 6, 0.0)
        def resetJointState(self, robot, i, v, *a): self.state[i] = v
        def getBasePositionAndOrientation(self, robot): return (tuple(self.base), (0.0, 0.0, 0.0, 1.0))
        def getLinkState(self, robot, link):
            arm = {27: 0, 45: 1}[link]
            act = [13, 14, 15, 16, 17, 19, 20] if arm == 0 else [31, 32, 33, 34, 35, 37, 38]
            pos, R, _, _ = OIK.fk(cb, self.state[act], arm)
            return (tuple(pos + self.base), tuple(RT.mat2quat(R.astype(np.float32))))
        def calculateInverseKinematics(self, robot, link, pos, **kw):
            self.calls.append(dict(link=link, pos=np.array(pos, dtype=float), orn=np.array(kw["targetOrientation"], dtype=float), rest=np.array(kw["restPoses"], dtype=float),
                                   lower=np.array(kw["lowerLimits"]), upper=np.array(kw["upperLimits"]), damping=np.array(kw["jointDamping"])))
            return list(self.answers[0 if link == 27 else 1])

    fbb = FakeBaxterBullet()
    BIK.p = fbb
    jb = [cb.arm_initqpos.copy()]
    bctl = BIK.BaxterIKController(bullet_data_path="/x", robot_jpos_getter=lambda: jb[0])
    out["bik_urdf"], out["bik_base"] = np.array(fbb.urdf), fbb.base
    out["bik_sync_target"] = np.array([bctl.ik_robot_target_pos_right, bctl.ik_robot_target_pos_left], dtype=float)
    out["bik_lower"], out["bik_upper"] = np.array(bctl.lower), np.array(bctl.upper)
    brec = dict(q=[], dpos=[], rot=[], answer=[], target_after=[], call_pos=[], call_orn=[], call_rest=[], ncalls=[], vel=[], links=[])
    for t in range(8):
        jb[0] = cb.arm_initqpos + rng.uniform(-0.3, 0.3, 14)
        dpos = rng.uniform(-0.1, 0.1, (2, 3))
        rot = np.stack([rand_rot(rng).astype(np.float32) for _ in range(2)])
        ans = rng.uniform(-0.5, 0.5, (2, 15))
        fbb.answers, fbb.calls = ans, []
        vel = bctl.get_control(right=dict(dpos=dpos[0].copy(), rotation=rot[0].copy()), left=dict(dpos=dpos[1].copy(), rotation=rot[1].copy()))
        brec["q"].append(jb[0].copy()); brec["dpos"].append(dpos); brec["rot"].append(rot.astype(float)); brec["answer"].append(ans)
        brec["target_after"].append(np.array([bctl.ik_robot_target_pos_right, bctl.ik_robot_target_pos_left], dtype=float))
        brec["call_pos"].append([fbb.calls[0]["pos"], fbb.calls[1]["pos"]]); brec["call_orn"].append([fbb.calls[0]["orn"], fbb.calls[1]["orn"]])
        brec["call_rest"].append([fbb.calls[0]["rest"], fbb.calls[1]["rest"]]); brec["ncalls"].append(len(fbb.calls)); brec["vel"].append(np.array(vel))
        brec["links"].append([fbb.calls[0]["link"], fbb.calls[1]["link"]])
    for k, v in brec.items():
        out["bik_" + k] = np.array(v)
    out["bik_user_sensitivity"] = np.array(bctl.user_sensitivity)
    out["bik_actual"] = np.array(bctl.actual)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB", len(out), "arrays")


if __name__ == "__main__":
    main()

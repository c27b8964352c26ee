"""Physics against MuJoCo's own recording of a manipulation: demos/Sawyer_7.pkl (tests/golden/demo_sawyer7.npz, 460 frames of a
Sawyer assembling swivel_chair_0700; scripts/make_golden_demo_sawyer7.py).  The reference replays such files state by state
(furniture.py:2183-2215); here the ROBOT is replayed and the PARTS are simulated:

  the arm and finger joints follow the recording kinematically -- their dofs get an armature of 1e5 (a prescribed-motion body: no
  force moves it) and, once per frame, the recorded joint positions and the constant velocity that reaches the next frame's;
  a frame is one env step of the IK-controlled env, 3 x 50 physics substeps -- and everything the parts do comes out of collision,
  soft contacts, friction and the Newton solver, compared with what they did in MuJoCo:

  * frames 0-47   the column hangs in the closed gripper (4 frames: 0.3 mm), the gripper opens, the column falls 8 cm and lands within
                  one frame (4 mm, orientation 0.04), then rests at MuJoCo's 14.9 mm for 40 frames while the arm moves about;
  * frames 255-428  the gripper closes on the seat (finger penetration 0.3-1 mm, MuJoCo's 20 N grip), lifts it 35 cm and carries it
                  for 150 frames = 45 s = 22 500 substeps: the seat stays within 1.3 cm of the recording (slip grows 0.06 mm per
                  frame) and within 0.05 in orientation.
  The RELEASE (frame 4), explained in round 4 (scripts/dev/release_diag.py): the recording holds one sample per
  frame, so the replay opens the fingers at constant speed over the frame's 150 substeps.  The pads' two remaining contacts then have
  -b v_n (opening speed) outweigh -k d r (penetration) from 1.5 mm of penetration on.  MuJoCo's default solver -- Newton on the primal
  problem, which is what the device implements and what `OracleSim.set_solver(kind="newton")` runs -- keeps the elliptic cone's
  middle-zone force there (normal residual satisfied, tangential not: the friction still carries 75-80 % of the column's weight)
  until the geometric contact ends at substep 48; the PGS solver (the oracle's other kind, which rounds 2-3 replayed with) updates the
  normal first and drops the friction with it at substep 37, so its column falls 11 substeps earlier and has landed when the frame
  ends -- like the recording, whose real fingers opened faster than the interpolation.  Substep by substep from the same states the
  device and the Newton oracle agree (column acceleration -1.751 vs -1.752, -2.071 vs -2.071, -3.027 vs -3.027 ...): the looser
  frame-4 tolerance belongs to the solver KIND (primal vs dual), not to fp32, and the fp64 Newton oracle needs it too.
  * frames 150-255 (round 4)  after the first connect: base and column are one welded body (weld 0 active, the recorded poses are the
                  aligned ones) standing on the floor while the arm travels to the seat and brushes it -- the welded pair stays within
                  0.1 mm of the recording for 105 frames = 15 750 substeps, the seat within 1.1 mm / 0.004.
  Not replayable, and why: frames 47-150 -- the arm pushes the lying column, which rolls (neutral equilibrium: any difference grows),
  and today's column collider is ~1 cm wider across the recorded grasp than the recorded finger opening allows (asset revision: the
  kinematic fingers end up 5-9 mm inside it); the connects of frames 148 / 430 are env logic (covered by tests/golden/env_logic.npz).
"""
import os

import numpy as np
import pytest

from furniture_amd.mjcf.model import CompiledModel, load_compiled

D = np.load(os.path.join(os.path.dirname(__file__), "golden", "demo_sawyer7.npz"))
N_SUB, H = 150, 0.002
SEGMENTS = {"hold_drop_rest": (0, 47), "grasp_lift_carry_seat": (255, 428), "welded_base_and_column_under_the_arm": (150, 255)}
WELDS = {"welded_base_and_column_under_the_arm": [0]}  # equality constraints that are active in a segment (0: base - column, connected at frame 148)


def kinematic_robot_model():
    m = load_compiled("Sawyer", "swivel_chair_0700")
    assert list(m.meta["part_names"]) == [str(x) for x in D["part_names"]]
    arr = dict(m.arrays)
    arr["dof_armature"] = np.array(arr["dof_armature"], dtype=np.float64).copy()
    arr["dof_armature"][:9] += 1e5  # arm 0..6, fingers 7, 8 (arm_dofadr, grip_dofadr)
    assert list(m.arm_dofadr) + list(m.grip_dofadr) == list(range(9))
    return CompiledModel(arr, m.meta)


def robot(t):
    return np.concatenate([D["arm"][t], D["grip"][t]])


def start_state(m, t):
    q = np.asarray(m.qpos0, dtype=np.float64).copy()
    q[:9] = robot(t)
    for i in range(m.nparts):
        a0 = int(m.part_qposadr[i])
        q[a0:a0 + 7] = D["parts"][t, i]
    return q


def errors(parts_sim, t):
    """per part: position error (m) and quaternion error (up to sign) against recorded frame t"""
    rec = D["parts"][t]
    dp = np.abs(parts_sim[:, :3] - rec[:, :3]).max(axis=1)
    dq = np.minimum(np.abs(parts_sim[:, 3:] - rec[:, 3:]), np.abs(parts_sim[:, 3:] + rec[:, 3:])).max(axis=1)
    return dp, dq


def check_segment(name, traj, slip=6e-3, primal=False):
    """traj[k] = part poses [nparts, 7] at the end of frame f0 + k + 1.  primal: the replay ran Newton on the primal problem (the device,
    the Newton oracle) -- the column leaves the linearly opening fingers 11 substeps later than under PGS (module docstring)"""
    f0, f1 = SEGMENTS[name]
    E = [errors(p, f0 + k + 1) for k, p in enumerate(traj)]
    dp, dq = np.array([e[0] for e in E]), np.array([e[1] for e in E])
    if name == "hold_drop_rest":
        BASE, COL, SEAT = 0, 1, 2
        assert dp[:4, COL].max() < 5e-4 and dq[:4, COL].max() < 0.01          # held in the closed gripper
        assert abs(D["parts"][4, COL, 2] - 0.0948) < 1e-3 and abs(D["parts"][5, COL, 2] - 0.0149) < 1e-4  # (the recording: released, landed one frame later)
        # fell 8 cm and came to rest within the frame (primal: the pads' friction holds until the contact ends, the column is still
        # 1.5 cm up when the frame ends; it lies where the recording has it one frame later)
        assert dp[4, COL] < (2.5e-2 if primal else 4e-3) and dq[4, COL] < (0.15 if primal else 0.04)  # (primal: caught in mid-fall, tumbling)
        assert dp[5:, COL].max() < (6e-3 if primal else 4e-3) and dq[5:, COL].max() < (0.06 if primal else 0.045)  # ... and stays there (a lying column is free to roll: 2.6 deg; where it comes to rest after the later tumble differs by 1-4 mm)
        z = np.array([p[COL, 2] for p in traj[5:]])
        assert np.abs(z - 0.0149).max() < 1e-4                                 # MuJoCo's resting height of the lying column
        assert dp[:, BASE].max() < 5e-4 and dp[:, SEAT].max() < 1e-3           # nothing else moves (the arm brushes the seat: 0.5 mm)
    elif name == "welded_base_and_column_under_the_arm":
        assert dp[:, :2].max() < 5e-4 and dq[:, :2].max() < 2e-3   # the welded base + column do not move
        assert dp[:, 2].max() < 2.5e-3 and dq[:, 2].max() < 0.01   # the arm brushes the seat on its way (1 mm)
    else:
        SEAT = 2
        assert dp[:60, SEAT].max() < 1.2e-2 and dp[25:60, SEAT].max() < slip   # closing, lift-off (the grip settles: 1 cm for three frames), lifting
        assert dp[:, SEAT].max() < 1.5e-2 and dq[:, SEAT].max() < 0.05         # 173 frames = 26 000 substeps in the gripper
        zs, zr = np.array([p[SEAT, 2] for p in traj]), D["parts"][f0 + 1:f1 + 1, SEAT, 2]
        assert zr.max() > 0.49 and abs(zs.max() - zr.max()) < 1e-2             # lifted 35 cm, as recorded
        assert dp[:, 0].max() < 2e-3                                           # the base is not disturbed


def replay_oracle(name, kind="newton"):
    from oracle.oracle_sim import OracleSim
    m = kinematic_robot_model()
    sim = OracleSim(m)
    sim.set_solver(100, 1e-10, kind)
    f0, f1 = SEGMENTS[name]
    sim.reset()
    sim.data.qpos[:] = start_state(m, f0)
    sim.data.qvel[:] = 0
    for e in WELDS.get(name, []):
        sim.model.eq_active[e] = 1
    sim.forward()
    traj = []
    for t in range(f0, f1):
        v = (robot(t + 1) - robot(t)) / (N_SUB * H)
        sim.data.qpos[:9], sim.data.qvel[:9] = robot(t), v
        sim.data.ctrl[:7], sim.data.ctrl[7:9] = v[:7], robot(t + 1)[7:9]  # (actuators at rest against the prescribed motion)
        for _ in range(N_SUB):
            sim.step()
        assert np.abs(sim.data.qpos[:9] - robot(t + 1)).max() < 2e-4  # the robot is where the recording has it
        traj.append(np.array([sim.data.qpos[int(a):int(a) + 7].copy() for a in m.part_qposadr]))
    sim.close()
    return traj


@pytest.mark.parametrize("name", sorted(SEGMENTS))
@pytest.mark.parametrize("kind", ["newton", "pgs"])
def test_oracle_parts_follow_the_mujoco_recording(name, kind):
    """both solver kinds of the fp64 oracle; Newton (MuJoCo's default, the device's algorithm) with the tolerances the device gets"""
    if kind == "pgs" and name in WELDS:
        pytest.skip("100 PGS sweeps do not converge on a weld row under load (the pair sags by millimetres); MuJoCo's default is Newton")
    check_segment(name, replay_oracle(name, kind), slip=8e-3 if kind == "newton" else 6e-3, primal=kind == "newton")


def test_the_release_differs_by_solver_kind_not_by_precision():
    """frame 4 in fp64 with both solvers: the primal solution still holds the column (friction of the two pad contacts) when PGS has
    dropped it -- 1.5 cm of height at the end of the frame, the whole of what rounds 2-3 booked as an fp32 effect"""
    zn = replay_oracle("hold_drop_rest", "newton")[4][1, 2]
    zp = replay_oracle("hold_drop_rest", "pgs")[4][1, 2]
    assert abs(zp - 0.0149) < 3e-3 and 0.025 < zn < 0.035, (zp, zn)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SEGMENTS))
def test_device_parts_follow_the_mujoco_recording(name):
    import torch
    from furniture_amd.sim import FSim, default_config
    m = kinematic_robot_model()
    sim = FSim(m, 2, config=default_config())
    f0, f1 = SEGMENTS[name]
    q = start_state(m, f0)
    zero = lambda k: np.zeros((1, k))
    sim.set_state(qpos=q[None], qvel=zero(m.nv), qacc_warmstart=zero(m.nv), ctrl=zero(m.nu), qfrc_applied=zero(m.nv), xfrc_applied=zero(6 * m.nparts))
    if WELDS.get(name):
        act = np.zeros((1, m.neq), dtype=np.int32)
        act[0, WELDS[name]] = 1
        sim.set_state(eq_active=act)
    traj = []
    for t in range(f0, f1):
        v = (robot(t + 1) - robot(t)) / (N_SUB * H)
        st = sim.get_state("qpos", "qvel")
        qp, qv = st["qpos"].clone(), st["qvel"].clone()
        qp[:, :9] = torch.as_tensor(robot(t), dtype=torch.float32, device=qp.device)
        qv[:, :9] = torch.as_tensor(v, dtype=torch.float32, device=qp.device)
        ctrl = np.concatenate([v[:7], robot(t + 1)[7:9]])
        sim.set_state(qpos=qp, qvel=qv, ctrl=ctrl[None])
        sim.physics_step(N_SUB)
        sim.sync()
        qn = sim.get_state("qpos")["qpos"]
        assert torch.equal(qn[0], qn[1])
        qn = qn[0].cpu().numpy().astype(np.float64)
        assert np.abs(qn[:9] - robot(t + 1)).max() < 5e-4
        traj.append(np.array([qn[int(a):int(a) + 7] for a in m.part_qposadr]))
    sim.close()
    check_segment(name, traj, slip=8e-3, primal=True)
    # and against the fp64 NEWTON oracle on the same protocol, frame by frame: held, released (the same substep: both are 3 cm up, in
    # mid-fall, when frame 4 ends), at rest
    ora = replay_oracle(name, "newton")
    d = np.array([np.abs(a[:, :3] - b[:, :3]).max() for a, b in zip(traj, ora)])
    if name == "hold_drop_rest":
        assert d[:4].max() < 2e-4 and d[4] < 4e-3 and d[5:].max() < 5e-3, (d[:4].max(), d[4], d[5:].max())
    elif name in WELDS:
        assert d.max() < 1.5e-3, d.max()
    else:
        assert d[:10].max() < 5e-4 and d.max() < 2e-2, (d[:10].max(), d.max())  # before the fingers touch; then two grips that settle and slip their own way

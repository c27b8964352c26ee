"""Capsule colliders (one in the in-scope assets, the column of Baxter's pedestal, robots/baxter/robot.xml:61).  Rounds 2-4
waived that geom by name on the device and skipped it in the oracle, so device == oracle said nothing about it.  Now both sides collide
it: its two end spheres against a plane (MuJoCo's mjc_PlaneCapsule), every other pair through the Minkowski-portal routine with the
capsule's support function (a sphere swept along a segment).  Scenario: a plank of desk_mikael_1064 is pushed 4 mm into the side of the
column and let go."""
import numpy as np
import pytest

from furniture_amd.mjcf.model import load_compiled
from furniture_amd.mjcf.reduce import WAIVED_COLLIDERS
from oracle.oracle_sim import OracleSim

PART, Y_TOUCH = 3, 0.446  # the plank whose long side faces the column, and the y at which that side reaches it


def _scene(m, y):
    q = m.qpos0.copy()
    q[m.arm_qposadr] = m.arm_initqpos
    q[m.grip_qposadr] = m.grip_initqpos
    for i in range(m.nparts):
        a = m.part_qposadr[i]
        q[a:a + 7] = m.part_initqpos[i]
    a = m.part_qposadr[PART]
    q[a + 1] = y
    q[a + 2] += 0.002
    return q


def _capsule_geom(m):
    A = m.arrays
    caps = np.where(A["cg_type"] == 3)[0]
    assert len(caps) == 1
    return int(caps[0]), int(A["cg_orig"][caps[0]])


def test_nothing_is_waived_and_the_pedestal_capsule_has_candidate_pairs():
    assert not WAIVED_COLLIDERS
    m = load_compiled("Baxter", "desk_mikael_1064")
    ci, go = _capsule_geom(m)
    assert m.meta["geom_names"][go] == "pedestal_2_collision"
    cp = m.arrays["cp"].reshape(-1, 3)
    mine = cp[(cp[:, 0] == ci) | (cp[:, 1] == ci)]
    assert len(mine) >= 16 and set(mine[:, 2]) == {10}  # PT_CONVEX: the portal routine


def test_oracle_collides_a_plank_with_the_pedestal_capsule():
    m = load_compiled("Baxter", "desk_mikael_1064")
    _, go = _capsule_geom(m)
    o = OracleSim(m)
    o.set_solver(100, 1e-10, "newton")
    # no contact a few millimetres short of the column, contact once the plank's side has reached it
    for y, want in ((Y_TOUCH - 0.006, False), (Y_TOUCH + 0.004, True)):
        o.reset()
        o.data.qpos[:] = _scene(m, y)
        o.forward()
        pairs = [(c, d) for c, d in zip(o.contacts(), o.contact_dists()) if go in c]
        assert bool(pairs) == want, (y, pairs)
    # the depth is the analytic one: column surface at 0.65 - 0.165, plank half width 0.04 (the plank lies on the floor, inside the
    # cylindrical part of the capsule, its long side perpendicular to the line to the axis: 5e-4 covers its 0.1 degree yaw)
    (c, d), = pairs
    assert abs(d - ((0.65 - 0.165) - (Y_TOUCH + 0.004 + 0.04))) < 5e-4, d
    # let go: the soft contact pushes the plank out and it comes to rest clear of the column
    rd = np.concatenate([m.arm_dofadr, m.grip_dofadr])
    o.data.qfrc_applied[rd] = o.data.qfrc_bias[rd]
    a = m.part_qposadr[PART]
    for _ in range(300):
        o.step()
    assert not [c for c in o.contacts() if go in c]
    assert Y_TOUCH - 0.004 < o.data.qpos[a + 1] < Y_TOUCH + 0.001 and abs(o.data.qvel[m.part_dofadr[PART] + 1]) < 1e-4


@pytest.mark.gpu
def test_device_matches_the_oracle_against_the_pedestal_capsule():
    import torch  # noqa: F401
    from furniture_amd.sim import FSim
    m = load_compiled("Baxter", "desk_mikael_1064")
    _, go = _capsule_geom(m)
    ys = [Y_TOUCH + 0.004, Y_TOUCH + 0.002, Y_TOUCH + 0.006, Y_TOUCH - 0.006]
    n = len(ys)
    q = np.stack([_scene(m, y) for y in ys])
    sim = FSim(m, n)
    sim.set_state(qpos=q, qvel=np.zeros((n, m.nv)), qacc_warmstart=np.zeros((n, m.nv)))
    sim.physics_forward()
    st = sim.get_state("qfrc_bias", "contact_geoms")
    bias = st["qfrc_bias"].cpu().numpy()
    rd = np.concatenate([m.arm_dofadr, m.grip_dofadr])
    app = np.zeros((n, m.nv))
    app[:, rd] = bias[:, rd]
    sim.set_state(qfrc_applied=app)
    oracles = []
    for e in range(n):
        o = OracleSim(m)
        o.set_solver(100, 1e-10, "newton")
        o.reset()
        o.data.qpos[:] = q[e]
        o.forward()
        o.data.qfrc_applied[rd] = o.data.qfrc_bias[rd]
        cg = st["contact_geoms"][e].cpu().numpy().reshape(-1, 2)
        dev = sorted(tuple(int(x) for x in r) for r in cg if r[0] >= 0)
        assert [c for c in dev if go in c] == sorted(c for c in o.contacts() if go in c), (e, dev)
        assert bool([c for c in dev if go in c]) == (e < 3)
        oracles.append(o)
    for block in range(3):  # 3 x 50 substeps: the push-out transient, then rest
        sim.physics_step(50)
        s2 = sim.get_state("qpos", "qvel")
        for e, o in enumerate(oracles):
            for _ in range(50):
                o.step()
            assert np.abs(s2["qpos"][e].cpu().numpy() - o.data.qpos).max() < 2e-5, (block, e)
            assert np.abs(s2["qvel"][e].cpu().numpy() - o.data.qvel).max() < 2e-3, (block, e)
    a = m.part_qposadr[PART]
    assert (s2["qpos"][:3, a + 1].cpu().numpy() < Y_TOUCH + 0.001).all()  # pushed out, on the device too
    sim.close()

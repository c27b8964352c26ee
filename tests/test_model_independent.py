"""Checks of the COMPILED MODEL that do not go through furniture_amd/mjcf/compile.py's own derivations (VERDICT r1 'weak' 1: the
oracle and the device share the model compiler, so a wrong invweight0 / inertia / friction rule is invisible to device-vs-oracle
tests).  Three independent angles:

1. MuJoCo-recorded equilibrium: frames of the reference's demos/Sawyer_7.pkl (recorded from MuJoCo) in which the swivel-chair
   parts lie at rest.  Resting heights depend on the whole chain geometry -> mass -> invweight0 -> solref/solimp impedance ->
   soft-contact penetration; today's compiled model must hold those poses (no penetration beyond 1 mm, drift below 0.2 mm over
   the 500 substeps between two recorded frames).
2. invweight0 by its DEFINITION (MuJoCo computation docs: dof_invweight0 = diagonal of M^-1 at qpos0, averaged over the
   translational / rotational dofs of a free joint; body_invweight0 = mean diagonal of the translational / rotational blocks of
   J M^-1 J' for the body's Jacobian at its inertial frame) evaluated with the oracle's dynamics by unit-force responses.
3. Mass and inertia of every part from a second, separate integration of the MJCF geoms (Monte-Carlo volume integration of the
   primitives; needs the reference's XML, so it only runs where /root/reference exists), and the friction / solmix rules on a
   two-geom contact."""
import os

import numpy as np
import pytest

from furniture_amd.mjcf.model import load_compiled
from oracle.oracle_sim import OracleSim

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _sim(m):
    o = OracleSim(m)
    o.set_solver(100, 1e-10, "newton")
    o.reset()
    return o


def test_mujoco_recorded_resting_poses_are_equilibria_of_the_compiled_model():
    m = load_compiled("Sawyer", "swivel_chair_0700")
    G = np.load(os.path.join(GOLD, "demo_static.npz"))
    names = [str(x) for x in G["part_names"]]
    order = [names.index(p) for p in m.meta["part_names"]]
    o = _sim(m)
    q = o.data.qpos
    q[m.arm_qposadr] = G["arm"][0]
    q[m.grip_qposadr] = G["grip"][0]
    for i, k in enumerate(order):
        a = m.part_qposadr[i]
        q[a:a + 7] = G["parts"][0, k]
    o.forward()
    rd = np.concatenate([m.arm_dofadr, m.grip_dofadr])
    o.data.qfrc_applied[rd] = o.data.qfrc_bias[rd]  # the demo's arm is position-held; here it is gravity-compensated in place
    floor = int(m.floor_geomid[0])
    part_geoms = set(int(g) for g in range(m.ngeom) if m.geom_is_partcol[g])
    pen = [d for (g1, g2), d in zip(o.contacts(), o.contact_dists()) if (g1 in part_geoms or g2 in part_geoms)]
    assert len(pen) >= 6 and floor in {g for c in o.contacts() for g in c}
    assert min(pen) > -1e-3, min(pen)  # MuJoCo's recorded resting poses sit in today's geometry without penetrating > 1 mm
    for _ in range(500):  # = the 10 recorded frames between frame 10 and frame 20
        o.step()
    for i, k in enumerate(order):
        a = m.part_qposadr[i]
        drift = np.abs(q[a:a + 3] - G["parts"][0, k, :3]).max()
        assert drift < 2e-4, (m.meta["part_names"][i], drift)
        assert np.abs(q[a:a + 3] - G["parts"][1, k, :3]).max() < 2e-4  # and agrees with what MuJoCo recorded 500 substeps later
        quat_err = min(np.abs(q[a + 3:a + 7] - G["parts"][1, k, 3:]).max(), np.abs(q[a + 3:a + 7] + G["parts"][1, k, 3:]).max())
        assert quat_err < 2e-3, (m.meta["part_names"][i], quat_err)
    o.close()


def _minv_columns(o, m):
    """columns of M^-1 at the current configuration from unit generalized-force responses (collisions off, no constraints)"""
    o.model.geom_contype[:] = 0
    o.model.geom_conaffinity[:] = 0
    o.forward()
    base = o.data.qacc.copy()
    cols = np.zeros((m.nv, m.nv))
    for d in range(m.nv):
        o.data.qfrc_applied[:] = 0
        o.data.qfrc_applied[d] = 1.0
        o.forward()
        cols[:, d] = o.data.qacc - base
    o.data.qfrc_applied[:] = 0
    return cols


@pytest.mark.parametrize("key", [("Sawyer", "table_lack_0825"), ("Sawyer", "swivel_chair_0700")])
def test_invweight0_equals_its_definition_through_the_oracle_dynamics(key):
    m = load_compiled(*key)
    o = _sim(m)  # qpos0
    Minv = _minv_columns(o, m)
    assert np.abs(Minv - Minv.T).max() < 1e-9 * np.abs(Minv).max()
    assert np.abs(Minv @ o.full_M() - np.eye(m.nv)).max() < 1e-8
    diag = np.diag(Minv)
    want = np.zeros(m.nv)
    for j in range(len(m.jnt_type)):
        d = int(m.jnt_dofadr[j])
        if int(m.jnt_type[j]) == 0:  # free joint: translational and rotational averages (MuJoCo's setInertia / set0)
            want[d:d + 3] = diag[d:d + 3].mean()
            want[d + 3:d + 6] = diag[d + 3:d + 6].mean()
        else:
            want[d] = diag[d]
    assert np.abs(m.dof_invweight0 / want - 1).max() < 1e-6, np.abs(m.dof_invweight0 / want - 1).max()
    # body_invweight0 of the moving bodies: mean diagonal of J M^-1 J' (translation at the inertial frame origin | rotation)
    checked = 0
    for b in range(1, m.nbody):
        jp, jr = o.body_jac(b, o.data.xipos[b])
        A = np.vstack([jp, jr]) @ Minv @ np.vstack([jp, jr]).T
        tr, ro = np.trace(A[:3, :3]) / 3, np.trace(A[3:, 3:]) / 3
        if tr < 1e-12:
            continue  # welded to the world
        assert abs(m.body_invweight0[b, 0] / tr - 1) < 1e-5 and abs(m.body_invweight0[b, 1] / ro - 1) < 1e-5, b
        checked += 1
    assert checked >= m.nparts + 7
    o.close()


def test_part_masses_and_inertias_by_a_second_integration_of_the_mjcf(have_reference):
    """mass, centre of mass and inertia tensor of every furniture part by Monte-Carlo integration of its MJCF primitives (uniform
    samples in each geom's bounding box, density from the XML) -- a derivation that shares no code with mjcf/compile.py."""
    if not have_reference:
        pytest.skip("needs the reference's MJCF assets")
    import xml.etree.ElementTree as ET
    from furniture_amd.mjcf.assemble import default_assets_root
    m = load_compiled("Sawyer", "table_lack_0825")
    root = ET.parse(os.path.join(default_assets_root(), "objects", "table_lack_0825.xml")).getroot()
    rng = np.random.RandomState(0)
    n = 400000
    seen = 0
    for body in root.iter("body"):
        name = body.get("name")
        if name not in m.meta["part_names"]:
            continue
        mass, first, second = 0.0, np.zeros(3), np.zeros((3, 3))
        for g in body.findall("geom"):
            if g.get("mass") is not None or float(g.get("density", "1000")) == 0 or g.get("type", "sphere") == "mesh":
                continue
            dens = float(g.get("density", "1000"))
            size = np.array([float(x) for x in g.get("size").split()])
            pos = np.array([float(x) for x in g.get("pos", "0 0 0").split()])
            assert g.get("euler") is None
            qw, qx, qy, qz = [float(x) for x in g.get("quat", "1 0 0 0").split()]
            R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                          [2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)],
                          [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)]])
            ty = g.get("type")
            if ty == "box":
                half = size[:3]
                pts = rng.uniform(-1, 1, (n // 2, 3)) * half
                pts = np.concatenate([pts, -pts])  # antithetic pairs: first moments of the symmetric primitives are exact
                inside = np.ones(n, dtype=bool)
            elif ty == "cylinder":
                half = np.array([size[0], size[0], size[1]])
                pts = rng.uniform(-1, 1, (n // 2, 3)) * half
                pts = np.concatenate([pts, -pts])
                inside = pts[:, 0] ** 2 + pts[:, 1] ** 2 <= size[0] ** 2
            else:
                pytest.fail("unexpected primitive %s" % ty)
            w = dens * np.prod(2 * half) / n
            p = pts[inside] @ R.T + pos
            mass += w * len(p)
            first += w * p.sum(axis=0)
            second += w * (p.T @ p)
        if mass == 0:
            continue
        com = first / mass
        I = np.eye(3) * np.trace(second) - second          # about the body origin
        I -= mass * (np.eye(3) * com @ com - np.outer(com, com))  # parallel axis to the centre of mass
        b = m.body_name2id(name)
        assert abs(m.body_mass[b] / mass - 1) < 5e-3, (name, m.body_mass[b], mass)
        assert np.abs(m.body_ipos[b] - com).max() < 2e-4 + 5e-3 * np.abs(com).max(), (name, m.body_ipos[b], com)
        ev = np.sort(np.linalg.eigvalsh(I))
        assert np.abs(np.sort(m.body_inertia[b]) / ev - 1).max() < 1e-2, (name, m.body_inertia[b], ev)
        seen += 1
    assert seen == m.nparts

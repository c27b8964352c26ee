"""Model compiler: sizes from SURVEY.md section 8, MJCF-derived masses, weld order, connector tables, blob layout."""
import struct

import numpy as np
import pytest

from furniture_amd.mjcf.model import load_compiled, BLOB_MAGIC

# SURVEY.md section 8 size table: nbody nq nv nu ngeom(colliding) nsite parts welds
SIZES = {
    ("Cursor", "toy_table"): (8, 35, 30, 0, 57, 52, 78, 5, 4),
    ("Sawyer", "table_lack_0825"): (36, 44, 39, 9, 51, 27, 86, 5, 4),
    ("Sawyer", "swivel_chair_0700"): (34, 30, 27, 9, 52, 30, 40, 3, 2),
    ("Baxter", "desk_mikael_1064"): (42, 47, 43, 18, 77, 38, 64, 4, 5),
    ("Sawyer", "toy_table"): (36, 44, 39, 9, 95, 71, 88, 5, 4),
}


@pytest.mark.parametrize("key", sorted(SIZES))
def test_sizes_match_survey(key):
    m = load_compiled(*key)
    nbody, nq, nv, nu, ngeom, ncol, nsite, nparts, neq = SIZES[key]
    assert (m.nbody, m.nq, m.nv, m.nu, m.ngeom, m.nsite, m.nparts, m.neq) == (nbody, nq, nv, nu, ngeom, nsite, nparts, neq)
    assert int(((m.geom_contype != 0) | (m.geom_conaffinity != 0)).sum()) == ncol


def test_table_lack_masses_and_order(sawyer_lack):
    m = sawyer_lack
    # legs: density 5 box 0.015 x 0.015 x 0.13125 -> 1.18 g ; top: density 50 box 0.32 x 0.12 x 0.02 -> 307 g
    mass = m.body_mass[m.part_bodyid]
    assert np.allclose(mass[:4], 5 * 8 * 0.015 * 0.015 * 0.13125, rtol=1e-12)
    assert np.isclose(mass[4], 50 * 8 * 0.32 * 0.12 * 0.02)
    assert m.meta["part_names"] == ["0_part0", "1_part1", "2_part2", "3_part3", "4_part4"]
    # weld order drives _get_next_subtask: (0,4),(2,4),(3,4),(1,4)  
    assert list(zip(m.eq_part1.tolist(), m.eq_part2.tolist())) == [(0, 4), (2, 4), (3, 4), (1, 4)]
    # actuator layout: 7 velocity actuators then r-finger, l-finger position actuators
    assert m.meta["actuator_names"][7:] == ["gripper_r_gripper_r_finger_joint", "gripper_r_gripper_l_finger_joint"]
    assert np.allclose(m.actuator_gain[:7], [8, 7, 6, 4, 2, 0.5, 0.1]) and np.allclose(m.actuator_gain[7:], 10000)
    # part colliders get friction 1 10 0.5, finger tips 2 10 0.5, floor 2 .005 .0001
    g = m.meta["geom_names"]
    assert np.allclose(m.geom_friction[g.index("noviz_collision_0_part0_0")], [1, 10, 0.5])
    assert np.allclose(m.geom_friction[g.index("l_fingertip_g0")], [2, 10, 0.5])
    assert np.allclose(m.geom_friction[g.index("FLOOR")], [2.0, 0.005, 0.0001])
    # connector tables: 4 leg sites (key leg-table) + 4 table sites (table-leg), angles 0/90/180/270
    assert m.conn_keya.tolist() == [0, 0, 0, 0, 1, 1, 1, 1] and m.conn_keyb.tolist() == [1, 1, 1, 1, 0, 0, 0, 0]
    assert np.allclose(m.conn_angles[:, :4], [0, 90, 180, 270]) and m.conn_nangle.tolist() == [4] * 8
    assert m.conn_partid.tolist() == [0, 1, 2, 3, 4, 4, 4, 4]


def test_part_order_is_document_order():
    m = load_compiled("Baxter", "desk_mikael_1064")
    assert m.meta["part_names"] == ["1_part1", "0_part0", "3_part3", "2_part2"]
    m = load_compiled("Sawyer", "toy_table")
    assert m.meta["part_names"] == ["0_part0", "4_part4", "3_part3", "2_part2", "1_part1"]


def test_reduced_model_preserves_mass(sawyer_lack):
    m = sawyer_lack
    moving = m.body_weldid != 0
    assert np.isclose(m.r_mass.sum(), m.body_mass[moving].sum())
    assert m.rdims.tolist()[:3] == [15, 6, 27]
    assert int(m.r_depth.max()) == 8


def test_blob_roundtrip(sawyer_lack):
    blob = sawyer_lack.to_blob()
    assert blob[:8] == BLOB_MAGIC
    ver, n = struct.unpack("<ii", blob[8:16])
    derived = {"cg_cursor", "cursor_pos0", "cg_namepart"}  # Cursor-agent tables derived from names at blob time
    assert n == len(sawyer_lack.arrays) + len(derived)
    found = {}
    for i in range(n):
        name, code, _, count, off = struct.unpack("<48siiqq", blob[16 + i * 72: 16 + (i + 1) * 72])
        name = name.rstrip(b"\0").decode()
        dt = "<f8" if code == 0 else "<i4"
        found[name] = np.frombuffer(blob, dtype=dt, count=count, offset=off)
    for k, v in sawyer_lack.arrays.items():
        assert np.array_equal(found[k], np.asarray(v).reshape(-1).astype(found[k].dtype)), k
    assert derived <= set(found) and not found["cg_cursor"].any()  # no cursor geoms in a Sawyer scene


def test_shipped_tables_match_fresh_compile(have_reference):
    if not have_reference:
        pytest.skip("reference MJCF assets not available on this machine")
    from furniture_amd.mjcf.model import build_model
    fresh = build_model("Sawyer", "table_lack_0825")
    shipped = load_compiled("Sawyer", "table_lack_0825")
    assert set(fresh.arrays) == set(shipped.arrays)
    for k in fresh.arrays:
        assert np.array_equal(np.asarray(fresh.arrays[k]), np.asarray(shipped.arrays[k])), k


def test_unsupported_collider_pairs_fail_the_compilation(have_reference, monkeypatch):
    """A primitive pair without a narrow-phase routine fails the model compilation -- never dropped silently, and since round 5 nothing is
    waived by name any more (Baxter's pedestal capsule, robots/baxter/robot.xml:61, collides: tests/test_capsule.py).  What is left
    without a routine: convex-mesh colliders (three furniture) and ellipsoids (none in the assets)."""
    if not have_reference:
        pytest.skip("needs the reference's MJCF assets")
    from furniture_amd.mjcf import reduce
    from furniture_amd.mjcf.model import build_model
    assert not reduce.WAIVED_COLLIDERS
    m = build_model("Baxter", "desk_mikael_1064")
    assert (m.arrays["cp"].reshape(-1, 3)[:, 2] == reduce.PT_CONVEX).sum() >= 16  # the capsule's pairs are in the candidate list
    monkeypatch.delitem(reduce._PAIR_CODE, (reduce.GEOM_CAPSULE, reduce.GEOM_BOX))
    with pytest.raises(NotImplementedError, match="pedestal_2_collision"):
        build_model("Baxter", "desk_mikael_1064")

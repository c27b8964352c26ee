"""Quaternion helpers vs golden vectors generated from the reference's transform_utils.py
(scripts/make_golden_transform.py) and its docstring known-answers."""
import os

import numpy as np

from furniture_amd import transform_utils as T

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "transform_utils.npz"))


def test_doc_known_answers():
    # transform_utils.py:35-36 and :703-712
    assert np.allclose(T.quat_multiply([1, -2, 3, 4], [-5, 6, 7, 8]), [-44, -14, 48, 28])
    assert np.allclose(G["doc_quat_multiply"], [-44, -14, 48, 28])
    assert np.allclose([T.angle_between((1, 0, 0), (0, 1, 0)), T.angle_between((1, 0, 0), (1, 0, 0)), T.angle_between((1, 0, 0), (-1, 0, 0))],
                       [1.5707963267948966, 0.0, 3.141592653589793])


def test_against_reference_vectors():
    q0, q1, v0, v1 = G["q0"], G["q1"], G["v0"], G["v1"]
    for i in range(len(q0)):
        assert np.array_equal(T.quat_multiply(q0[i], q1[i]), G["quat_multiply"][i])
        assert np.allclose(T.quat_slerp(q0[i], q1[i], G["slerp_frac"][i]), G["quat_slerp"][i], atol=1e-6)
        assert np.allclose(T.lookat_to_quat(v0[i], v1[i]), G["lookat_to_quat"][i], atol=1e-12)
        assert np.allclose(T.rotate_vector(v0[i], v1[i], G["angles"][i]), G["rotate_vector"][i], atol=1e-12)
        assert np.allclose(T.rotate_vector_cos_siml(v0[i], v1[i], G["cos"][i], 1), G["rotate_vector_cos_siml_pos"][i], atol=1e-12)
        assert np.allclose(T.rotate_vector_cos_siml(v0[i], v1[i], G["cos"][i], -1), G["rotate_vector_cos_siml_neg"][i], atol=1e-12)
        assert abs(T.cos_siml(v0[i], v1[i]) - G["cos_siml"][i]) < 1e-14
        assert np.array_equal(T.unit_vector(v0[i]), G["unit_vector"][i])
        assert abs(T.angle_between(v0[i], v1[i]) - G["angle_between"][i]) < 1e-6
        assert np.array_equal(T.convert_quat(q0[i], "xyzw"), G["convert_xyzw"][i])
        a, b = T.mat2quat(G["rotmats"][i]), G["mat2quat"][i]
        assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 1e-6


def test_quaternion_class_semantics():
    rng = np.random.RandomState(0)
    from scipy.spatial.transform import Rotation as R
    for _ in range(20):
        q = rng.randn(4) * rng.uniform(0.5, 2)
        v = rng.randn(3)
        qq = T.Quaternion(q)
        ref = R.from_quat([q[1], q[2], q[3], q[0]]).apply(v)  # scipy normalises, like pyquaternion.rotate
        assert np.allclose(qq.rotate(v), ref, atol=1e-12)
        assert np.allclose((qq * qq.inverse).q, [1, 0, 0, 0], atol=1e-12)
    # euler_to_quat = qz*qy*qx (degrees), rel_pose / transform_to_target_quat round trip
    e = T.euler_to_quat([10, 20, 30])
    ref = R.from_euler("xyz", [10, 20, 30], degrees=True).as_quat()
    assert np.allclose(e, [ref[3], ref[0], ref[1], ref[2]], atol=1e-12)
    p1 = np.array([0.1, 0.2, 0.3, *T.euler_to_quat([5, 6, 7])])
    p2 = np.array([-0.3, 0.1, 0.5, *T.euler_to_quat([50, -16, 70])])
    rel = T.rel_pose(p1, p2)
    back = T.Quaternion(p1[3:]).rotate(rel[:3]) + p1[:3]
    assert np.allclose(back, p2[:3], atol=1e-12)
    assert np.allclose((T.Quaternion(p1[3:]) * T.Quaternion(rel[3:])).q, p2[3:], atol=1e-12)
    npos, nq = T.transform_to_target_quat(p1, p2, T.euler_to_quat([1, 2, 3]))
    # rigid motion: relative pose is preserved
    rel2 = T.rel_pose(np.concatenate([p1[:3], T.euler_to_quat([1, 2, 3])]), np.concatenate([npos, nq]))
    assert np.allclose(rel2, rel, atol=1e-12)

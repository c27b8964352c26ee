"""Generate tests/golden/transform_utils.npz by IMPORTING the reference's transform_utils.py.

Runs only in the build container (needs /root/reference).  pyquaternion is absent here, so a stub module is
installed for the import; only functions that never touch pyquaternion are exercised
(quat_multiply, quat_slerp, lookat_to_quat, rotate_vector, rotate_vector_cos_siml, cos_siml, unit_vector,
convert_quat, mat2quat, angle_between), plus the docstring known-answers (transform_utils.py:35-36, 703-712).
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/furniture/env/transform_utils.py"
stub = types.ModuleType("pyquaternion")
stub.Quaternion = object
sys.modules["pyquaternion"] = stub
spec = importlib.util.spec_from_file_location("ref_transform_utils", REF)
R = importlib.util.module_from_spec(spec)
spec.loader.exec_module(R)

rng = np.random.RandomState(7)
n = 64
out = {}
q0 = rng.randn(n, 4); q1 = rng.randn(n, 4)
out["q0"], out["q1"] = q0, q1
out["quat_multiply"] = np.stack([R.quat_multiply(a, b) for a, b in zip(q0, q1)])
fr = rng.uniform(0.05, 0.95, n)
out["slerp_frac"] = fr
out["quat_slerp"] = np.stack([R.quat_slerp(a, b, f) for a, b, f in zip(q0, q1, fr)])
v0 = rng.randn(n, 3); v1 = rng.randn(n, 3)
out["v0"], out["v1"] = v0, v1
out["lookat_to_quat"] = np.stack([R.lookat_to_quat(a, b) for a, b in zip(v0, v1)])
ang = rng.uniform(-360, 360, n)
out["angles"] = ang
out["rotate_vector"] = np.stack([R.rotate_vector(a, b, t) for a, b, t in zip(v0, v1, ang)])
cs = rng.uniform(-1, 1, n)
out["cos"] = cs
out["rotate_vector_cos_siml_pos"] = np.stack([R.rotate_vector_cos_siml(a, b, c, 1) for a, b, c in zip(v0, v1, cs)])
out["rotate_vector_cos_siml_neg"] = np.stack([R.rotate_vector_cos_siml(a, b, c, -1) for a, b, c in zip(v0, v1, cs)])
out["cos_siml"] = np.array([R.cos_siml(a, b) for a, b in zip(v0, v1)])
out["unit_vector"] = np.stack([R.unit_vector(a) for a in v0])
out["angle_between"] = np.array([R.angle_between(a, b) for a, b in zip(v0, v1)])
out["convert_xyzw"] = np.stack([R.convert_quat(a, "xyzw") for a in q0])
# rotation matrices from random unit quaternions (xyzw) via the reference's own quat2mat
qs = q0 / np.linalg.norm(q0, axis=1, keepdims=True)
mats = np.stack([R.quat2mat(q) for q in qs])
out["rotmats"] = mats
out["mat2quat"] = np.stack([R.mat2quat(m) for m in mats])
out["doc_quat_multiply"] = R.quat_multiply([1, -2, 3, 4], [-5, 6, 7, 8])
out["doc_angle_between"] = np.array([R.angle_between((1, 0, 0), (0, 1, 0)), R.angle_between((1, 0, 0), (1, 0, 0)), R.angle_between((1, 0, 0), (-1, 0, 0))])
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "transform_utils.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, {k: np.shape(v) for k, v in out.items()})

"""Quaternion / frame helpers used by the connector state machine.

Host-side (numpy, float64) restatement of the subset of the reference's
``furniture/env/transform_utils.py`` that FurnitureEnv's hot path calls
(reference lines cited per function).  ``pyquaternion`` is not available in
this image, so a tiny Hamilton ``Quaternion`` (w, x, y, z) with the semantics
the reference relies on is provided: ``rotate()`` normalises first and
``inverse`` is conj / |q|^2.

The same formulas are implemented again on the device in
``csrc/fsim_connect.hpp``; tests compare the two.
"""

import math

import numpy as np

_EPS = np.finfo(float).eps * 4.0


class Quaternion:
    """Hamilton quaternion, components ordered (w, x, y, z)."""

    __slots__ = ("q",)

    def __init__(self, *args, axis=None, degrees=None, radians=None):
        if axis is not None:
            ang = math.radians(degrees) if degrees is not None else float(radians)
            ax = np.asarray(axis, dtype=np.float64)
            ax = ax / np.linalg.norm(ax)
            s = math.sin(0.5 * ang)
            self.q = np.array([math.cos(0.5 * ang), ax[0] * s, ax[1] * s, ax[2] * s])
        elif len(args) == 0:
            self.q = np.array([1.0, 0.0, 0.0, 0.0])
        elif len(args) == 1:
            a = args[0]
            self.q = np.array(a.q if isinstance(a, Quaternion) else a, dtype=np.float64)
        else:
            self.q = np.array(args, dtype=np.float64)
        assert self.q.shape == (4,)

    # -- algebra ---------------------------------------------------------
    def __mul__(self, other):
        a, b = self.q, Quaternion(other).q
        return Quaternion(
            a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
            a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
            a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
        )

    @property
    def conjugate(self):
        return Quaternion(self.q[0], -self.q[1], -self.q[2], -self.q[3])

    @property
    def inverse(self):
        n2 = float(np.dot(self.q, self.q))
        return Quaternion(self.conjugate.q / n2)

    @property
    def normalised(self):
        return Quaternion(self.q / np.linalg.norm(self.q))

    def rotate(self, v):
        """Rotate a 3-vector; like pyquaternion the quaternion is normalised first."""
        u = self.normalised
        p = Quaternion(0.0, v[0], v[1], v[2])
        return (u * p * u.conjugate).q[1:].copy()

    @property
    def rotation_matrix(self):
        return quat2mat_wxyz(self.normalised.q)

    def __iter__(self):
        return iter(self.q)

    def __getitem__(self, i):
        return self.q[i]

    def __repr__(self):
        return "Quaternion(%r)" % (self.q.tolist(),)


def quat2mat_wxyz(q):
    w, x, y, z = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def convert_quat(q, to="xyzw"):
    """ref transform_utils.py:15-30."""
    q = np.asarray(q)
    if to == "xyzw":
        return q[[1, 2, 3, 0]]
    if to == "wxyz":
        return q[[3, 0, 1, 2]]
    raise ValueError("convert_quat: `to` must be 'xyzw' or 'wxyz'")


def quat_multiply(q1, q0):
    """xyzw product q1*q0, float32 result (ref transform_utils.py:33-50)."""
    x0, y0, z0, w0 = q0
    x1, y1, z1, w1 = q1
    return np.array(
        [
            x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
            -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
            x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0,
            -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0,
        ],
        dtype=np.float32,
    )


def unit_vector(v):
    """float32 normalisation of a 1-D vector (ref transform_utils.py:53-97)."""
    d = np.array(v, dtype=np.float32, copy=True)
    if d.size == 0:
        return d
    d /= math.sqrt(float(np.dot(d, d)))
    return d


def quat_slerp(q0, q1, fraction, spin=0, shortestpath=True):
    """ref transform_utils.py:122-160 (operates on unit_vector'ed float32 copies)."""
    a = unit_vector(np.asarray(q0)[:4])
    b = unit_vector(np.asarray(q1)[:4])
    if fraction == 0.0:
        return a
    if fraction == 1.0:
        return b
    d = float(np.dot(a, b))
    if abs(abs(d) - 1.0) < _EPS:
        return a
    if shortestpath and d < 0.0:
        d = -d
        b *= -1.0
    d = min(d, 1.0)
    ang = math.acos(d) + spin * math.pi
    if abs(ang) < _EPS:
        return a
    isin = 1.0 / math.sin(ang)
    a *= math.sin((1.0 - fraction) * ang) * isin
    b *= math.sin(fraction * ang) * isin
    a += b
    return a


def _norm(x):
    return x / np.linalg.norm(x)


def lookat_to_quat(forward, up):
    """xyzw quaternion of the frame (x=up x fwd, y=fwd x x, z=fwd).

    ref transform_utils.py:457-512.  Note the reference calls it with
    (up-vector, forward-vector), i.e. the site z-axis is passed as ``forward``.
    """
    f = _norm(np.asarray(forward, dtype=np.float64))
    s = _norm(np.cross(_norm(np.asarray(up, dtype=np.float64)), f))
    u = np.cross(f, s)
    m00, m01, m02 = s
    m10, m11, m12 = u
    m20, m21, m22 = f
    tr = (m00 + m11) + m22
    q = np.zeros(4)
    if tr > 0:
        n = math.sqrt(tr + 1)
        q[3] = n * 0.5
        n = 0.5 / n
        q[0] = (m12 - m21) * n
        q[1] = (m20 - m02) * n
        q[2] = (m01 - m10) * n
    elif m00 >= m11 and m00 >= m22:
        n = math.sqrt(((1 + m00) - m11) - m22)
        k = 0.5 / n
        q[:] = [0.5 * n, (m01 + m10) * k, (m02 + m20) * k, (m12 - m21) * k]
    elif m11 > m22:
        n = math.sqrt(((1 + m11) - m00) - m22)
        k = 0.5 / n
        q[:] = [(m10 + m01) * k, 0.5 * n, (m21 + m12) * k, (m20 - m02) * k]
    else:
        n = math.sqrt(((1 + m22) - m00) - m11)
        k = 0.5 / n
        q[:] = [(m20 + m02) * k, (m21 + m12) * k, 0.5 * n, (m01 - m10) * k]
    return q


def euler_to_quat(rotation_deg, quat=None):
    """wxyz list; q = qz*qy*qx, optionally left-multiplied by ``quat``
    (ref transform_utils.py:617-630)."""
    qx = Quaternion(axis=[1, 0, 0], degrees=rotation_deg[0])
    qy = Quaternion(axis=[0, 1, 0], degrees=rotation_deg[1])
    qz = Quaternion(axis=[0, 0, 1], degrees=rotation_deg[2])
    q = qz * qy * qx
    if quat is not None:
        q = Quaternion(quat) * q
    return list(q)


def rel_pose(qpos1, qpos2):
    """Pose of qpos2 in qpos1's frame: [R1^-1 (p2-p1), q1^-1 q2] (ref :633-638)."""
    q1inv = Quaternion(qpos1[3:7]).inverse
    rq = q1inv * Quaternion(qpos2[3:7])
    rp = q1inv.rotate(np.asarray(qpos2[:3]) - np.asarray(qpos1[:3]))
    return np.concatenate([rp, rq.q])


def transform_to_target_quat(qpos_base, qpos, target_quat):
    """Pose of ``qpos`` after rigidly rotating ``qpos_base`` to ``target_quat``
    about the base position (ref :641-664)."""
    base_p = np.asarray(qpos_base[:3], dtype=np.float64)
    rel = Quaternion(target_quat) * Quaternion(qpos_base[3:7]).inverse
    new_p = rel.rotate(np.asarray(qpos[:3], dtype=np.float64) - base_p) + base_p
    new_q = rel * Quaternion(qpos[3:7])
    return new_p, list(new_q)


def l2_dist(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)))


def cos_siml(a, b):
    """ref :718-720."""
    return float(np.dot(a, b) / np.linalg.norm(a) / np.linalg.norm(b))


def rotate_vector(v, axis, angle_deg):
    """cos(a) v + sin(a) k x v  -- note: no (1-cos)(k.v)k term, as in ref :739-745."""
    v = np.asarray(v)
    k = unit_vector(axis)
    a = angle_deg / 180 * math.pi
    return np.cos(a) * v + np.sin(a) * np.cross(k, v)


def rotate_vector_cos_siml(v, axis, cos, direction):
    """ref :748-754."""
    assert direction in (-1, 1)
    v = np.asarray(v)
    k = unit_vector(axis)
    return cos * v + direction * np.sqrt(1 - cos ** 2) * np.cross(k, v)


def angle_between(v1, v2):
    """ref :699-715."""
    a = unit_vector(v1)
    b = unit_vector(v2)
    return float(np.arccos(np.clip(np.dot(a, b), -1.0, 1.0)))


def mat2quat(rmat):
    """xyzw quaternion of a rotation matrix (largest-eigenvector method,
    ref :298-352, non-precise branch)."""
    M = np.asarray(rmat, dtype=np.float32)[:3, :3].astype(np.float64)
    m00, m01, m02 = M[0]
    m10, m11, m12 = M[1]
    m20, m21, m22 = M[2]
    K = np.array(
        [
            [m00 - m11 - m22, 0.0, 0.0, 0.0],
            [m01 + m10, m11 - m00 - m22, 0.0, 0.0],
            [m02 + m20, m12 + m21, m22 - m00 - m11, 0.0],
            [m21 - m12, m02 - m20, m10 - m01, m00 + m11 + m22],
        ]
    )
    K /= 3.0
    w, V = np.linalg.eigh(K)
    q = V[[3, 0, 1, 2], np.argmax(w)]
    if q[0] < 0.0:
        q = -q
    return q[[1, 2, 3, 0]]


def quat_inverse(q):
    """xyzw; conjugate / |q|^2 (ref :112-119; doctest: quat_multiply(q, quat_inverse(q)) == [0, 0, 0, 1])."""
    q = np.asarray(q)
    conj = np.array((-q[0], -q[1], -q[2], q[3]), dtype=np.float32)  # quat_conjugate down-casts to float32 (ref :99-109)
    return conj / np.dot(q, q)


def quat2mat(quaternion):
    """xyzw -> 3x3 (ref :207-229: float32 copy, normalised through sqrt(2 / n))."""
    q = np.array(quaternion, dtype=np.float32, copy=True)[[3, 0, 1, 2]]
    n = np.dot(q, q)
    if n < np.finfo(float).eps * 4.0:
        return np.identity(3)
    q *= math.sqrt(2.0 / n)
    q = np.outer(q, q)
    return np.array([[1.0 - q[2, 2] - q[3, 3], q[1, 2] - q[3, 0], q[1, 3] + q[2, 0]],
                     [q[1, 2] + q[3, 0], 1.0 - q[1, 1] - q[3, 3], q[2, 3] - q[1, 0]],
                     [q[1, 3] - q[2, 0], q[2, 3] + q[1, 0], 1.0 - q[1, 1] - q[2, 2]]])

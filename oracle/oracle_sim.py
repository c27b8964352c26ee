"""ctypes wrapper around libfsim_oracle.so shaped like the slice of ``mujoco_py.MjSim``
the reference env code touches.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Parity unpinned against MuJoCo itself (see
fsim_oracle.h).
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libfsim_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("fsim_oracle.c", "fsim_oracle.h", "fsim_oracle_collide.inc", "fsim_oracle_solve.inc")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(os.environ.get("OSIM_LIB") or build())  # (OSIM_LIB: the sanitizer build, tests/test_oracle_sanitizers.py)
        L.osim_create.restype = ctypes.c_void_p
        L.osim_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.osim_destroy.argtypes = [ctypes.c_void_p]
        L.osim_last_error.restype = ctypes.c_char_p
        L.osim_dptr.restype = ctypes.POINTER(ctypes.c_double)
        L.osim_dptr.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.osim_iptr.restype = ctypes.POINTER(ctypes.c_int32)
        L.osim_iptr.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        for f in ("osim_reset_data", "osim_forward"):
            getattr(L, f).argtypes = [ctypes.c_void_p]
            getattr(L, f).restype = None
        L.osim_step.argtypes = [ctypes.c_void_p]
        L.osim_step.restype = ctypes.c_int
        L.osim_site_vel.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.osim_body_jac.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.osim_full_M.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.osim_set_solver.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
        L.osim_set_solver_kind.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.osim_last_solver_iters.argtypes = [ctypes.c_void_p]
        L.osim_last_solver_iters.restype = ctypes.c_int
        L.osim_contact_dist.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.osim_contact_dist.restype = ctypes.c_double
        _LIB = L
    return _LIB


class OracleCapacity(RuntimeError):
    """a substep had more contacts / constraint rows than the checker holds"""


class SimUnstable(Exception):
    """Analogue of mujoco_py.MujocoException raised from sim.step()."""


class _Views:
    pass


class OracleSim:
    def __init__(self, model):
        """model: furniture_amd.mjcf.model.CompiledModel"""
        self.cm = model
        L = lib()
        blob = model.to_blob()
        self._h = L.osim_create(blob, len(blob))
        if not self._h:
            raise RuntimeError("osim_create: %s" % L.osim_last_error().decode())
        self.data = _Views()
        self.model = _Views()
        nb, ng, ns = model.nbody, model.ngeom, model.nsite
        shapes = dict(xpos=(nb, 3), xquat=(nb, 4), xmat=(nb, 9), xipos=(nb, 3), geom_xpos=(ng, 3), geom_xmat=(ng, 9),
                      site_xpos=(ns, 3), site_xmat=(ns, 9), xfrc_applied=(nb, 6), cvel=(nb, 6))
        for name in ("qpos", "qvel", "ctrl", "qfrc_applied", "xfrc_applied", "qacc", "qacc_warmstart", "qfrc_bias",
                     "qfrc_constraint", "qfrc_actuator", "qfrc_passive", "qacc_smooth", "actuator_force", "xpos", "xquat",
                     "xmat", "xipos", "geom_xpos", "geom_xmat", "site_xpos", "site_xmat", "time", "cvel"):
            setattr(self.data, name, self._dview(name, shapes.get(name)))
        self.data.body_xpos, self.data.body_xquat = self.data.xpos, self.data.xquat
        self.model.eq_data = self._dview("eq_data", (model.neq, 7))
        self.model.body_pos = self._dview("body_pos", (nb, 3))
        for name in ("geom_contype", "geom_conaffinity", "eq_active"):
            setattr(self.model, name, self._iview(name))
        self._cg1 = self._iview("contact_geom1")
        self._cg2 = self._iview("contact_geom2")
        self._ncon = self._iview("ncon")
        self._nefc = self._iview("nefc")
        self._ndropped = self._iview("ndropped")
        # MuJoCo's default solver (the reference sets none: base.xml:4 has impratio and cone only) is Newton on the primal problem; the C
        # struct's zero-initialised kind is PGS.  Round 4: a replay that forgot to say "newton" compared the device with PGS for two
        # rounds (tests/test_demo_sawyer_replay.py) -- the Python wrapper now starts where MuJoCo does; PGS is an explicit choice
        L.osim_set_solver_kind(self._h, 1)

    def _dview(self, name, shape=None):
        n = ctypes.c_int()
        p = lib().osim_dptr(self._h, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        a = np.ctypeslib.as_array(p, shape=(max(n.value, 0),)) if n.value else np.zeros(0)
        return a.reshape(shape) if shape is not None and n.value else a

    def _iview(self, name):
        n = ctypes.c_int()
        p = lib().osim_iptr(self._h, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(max(n.value, 0),)) if n.value else np.zeros(0, np.int32)

    # -- mujoco_py-ish surface -------------------------------------------
    def reset(self):
        lib().osim_reset_data(self._h)

    def forward(self):
        lib().osim_forward(self._h)

    def step(self):
        rc = lib().osim_step(self._h)
        if rc:
            raise SimUnstable("oracle step rc=%d" % rc)
        if self._ndropped[0]:  # the checker never clips silently (fsim_oracle.c MAXCON / MAXEFC)
            raise OracleCapacity("the oracle dropped %d contacts / rows: raise MAXCON / MAXEFC in oracle/fsim_oracle.c" % int(self._ndropped[0]))

    @property
    def ncon(self):
        return int(self._ncon[0])

    @property
    def nefc(self):
        return int(self._nefc[0])

    def contacts(self):
        n = self.ncon
        return list(zip(self._cg1[:n].tolist(), self._cg2[:n].tolist()))

    def contact_dists(self):
        """data.contact[i].dist of the listed contacts (< 0: penetration)"""
        return [lib().osim_contact_dist(self._h, i) for i in range(self.ncon)]

    def site_vel(self, site_id):
        """data.site_xvelp[site], data.site_xvelr[site] as mujoco_py computes them: jac(site) . qvel, i.e. the Jacobian
        of the last forward pass (one integration old after sim.step()) times the current qvel."""
        jp, jr = self.body_jac(int(self.cm.site_bodyid[site_id]), self.data.site_xpos[site_id])
        return jp @ self.data.qvel, jr @ self.data.qvel

    def site_vel_cvel(self, site_id):
        """mj_objectVelocity-style: from the body velocities (cvel) of the last forward pass."""
        vp, vr = np.zeros(3), np.zeros(3)
        lib().osim_site_vel(self._h, int(site_id), vp.ctypes.data, vr.ctypes.data)
        return vp, vr

    def body_jac(self, body, point):
        nv = self.cm.nv
        jp, jr = np.zeros((3, nv)), np.zeros((3, nv))
        pt = np.ascontiguousarray(point, dtype=np.float64)
        lib().osim_body_jac(self._h, int(body), pt.ctypes.data, jp.ctypes.data, jr.ctypes.data)
        return jp, jr

    def full_M(self):
        nv = self.cm.nv
        M = np.zeros((nv, nv))
        lib().osim_full_M(self._h, M.ctypes.data)
        return M

    def set_solver(self, iterations=100, tolerance=1e-8, kind=None):
        lib().osim_set_solver(self._h, int(iterations), float(tolerance))
        if kind is not None:
            lib().osim_set_solver_kind(self._h, {"pgs": 0, "newton": 1}[kind])

    @property
    def last_solver_iters(self):
        return lib().osim_last_solver_iters(self._h)

    def close(self):
        if self._h:
            lib().osim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

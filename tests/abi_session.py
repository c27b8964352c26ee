"""One ctypes session against the C-ABI of include/fsim.h, written ONCE and run against two libraries: furniture_amd/csrc/libfsim.so
(device pointers: torch is the allocator) and oracle/libfsim_cpu.so (host pointers: numpy).  Every call below is the same call with
the same arguments on both; only where the buffers live differs.  Test infrastructure."""

import ctypes
import os

import numpy as np

from furniture_amd.sim import FsimConfig, StatePtrs, INFO_DIM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB = os.path.join(ROOT, "oracle", "libfsim_cpu.so")
GPU_LIB = os.path.join(ROOT, "furniture_amd", "csrc", "libfsim.so")

_NP2T = {"float32": "float32", "uint8": "uint8", "int32": "int32"}


class Abi:
    """A loaded library + the memory its pointers refer to (``device``: None = host, else a torch device)."""

    def __init__(self, path, device=None):
        if device is not None:
            import torch  # noqa: F401 -- torch binds its HIP runtime first (furniture_amd/sim.py lib())
        self.L = L = ctypes.CDLL(path)
        self.device = device
        L.fsim_last_error.restype = ctypes.c_char_p
        L.fsim_kernel_variant.restype = ctypes.c_char_p
        L.fsim_kernel_variant.argtypes = [ctypes.c_void_p]
        L.fsim_default_config.argtypes = [ctypes.POINTER(FsimConfig)]
        L.fsim_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(FsimConfig), ctypes.POINTER(ctypes.c_void_p)]
        L.fsim_destroy.argtypes = [ctypes.c_void_p]
        L.fsim_destroy.restype = None
        L.fsim_dims.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int32)] * 7
        L.fsim_sync.argtypes = [ctypes.c_void_p]
        L.fsim_tables_needed.argtypes = [ctypes.c_void_p]
        L.fsim_set_reset_tables.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.fsim_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.fsim_step.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 5
        L.fsim_get_state.argtypes = [ctypes.c_void_p, ctypes.POINTER(StatePtrs)]
        L.fsim_set_state.argtypes = [ctypes.c_void_p, ctypes.POINTER(StatePtrs)]
        L.fsim_set_max_episode_steps.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.fsim_physics_forward.argtypes = [ctypes.c_void_p]
        L.fsim_set_dense_reward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.fsim_set_preassembled.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.fsim_set_init_state.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.fsim_set_attach_noise.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

    def check(self, rc):
        if rc != 0:
            raise RuntimeError((self.L.fsim_last_error() or b"").decode())

    # -- memory ----------------------------------------------------------------
    def zeros(self, shape, dtype):
        if self.device is None:
            return np.zeros(shape, dtype=dtype)
        import torch
        return torch.zeros(shape, dtype=getattr(torch, _NP2T[np.dtype(dtype).name]), device=self.device)

    def put(self, buf, arr):
        if self.device is None:
            buf[...] = arr
        else:
            import torch
            buf.copy_(torch.as_tensor(np.ascontiguousarray(arr)).to(buf.dtype))
            torch.cuda.synchronize()

    def get(self, buf):
        return buf.copy() if self.device is None else buf.cpu().numpy()

    def ptr(self, buf):
        if buf is None:
            return None
        return buf.ctypes.data if self.device is None else buf.data_ptr()


class Session:
    def __init__(self, abi, blob, n, **cfg_kw):
        self.abi, self.n = abi, n
        L = abi.L
        cfg = FsimConfig()
        L.fsim_default_config(ctypes.byref(cfg))
        for k, v in cfg_kw.items():
            setattr(cfg, k, v)
        self.h = ctypes.c_void_p()
        abi.check(L.fsim_create(blob, len(blob), n, 0, ctypes.byref(cfg), ctypes.byref(self.h)))
        d = [ctypes.c_int32() for _ in range(7)]
        abi.check(L.fsim_dims(self.h, *[ctypes.byref(x) for x in d]))
        self.nq, self.nv, self.nu, self.dof, self.obs_dim, self.info_dim, _ = [x.value for x in d]
        assert self.info_dim == INFO_DIM
        L.fsim_max_contacts.argtypes = [ctypes.c_void_p]
        self.max_contacts = L.fsim_max_contacts(self.h)
        self.obs = abi.zeros((n, self.obs_dim), np.float32)
        self.act = abi.zeros((n, self.dof), np.float32)
        self.rew = abi.zeros((n,), np.float32)
        self.done = abi.zeros((n,), np.uint8)
        self.info = abi.zeros((n, INFO_DIM), np.int32)

    def variant(self):
        return self.abi.L.fsim_kernel_variant(self.h).decode()

    def set_reset_tables(self, parts, noise, n_noise=101, mask=None):
        parts, noise = np.ascontiguousarray(parts, dtype=np.float32), np.ascontiguousarray(noise, dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.abi.check(self.abi.L.fsim_set_reset_tables(self.h, None if m is None else m.ctypes.data, parts.ctypes.data, noise.ctypes.data, n_noise))

    def set_dense_reward(self, coef, subtasks):
        """furniture_amd.dense.pack_dense(model) -> the tables of a dense_reward = 1 handle (host pointers on both libraries)"""
        coef, subtasks = np.ascontiguousarray(coef, dtype=np.float32), np.ascontiguousarray(subtasks, dtype=np.float32)
        self.abi.check(self.abi.L.fsim_set_dense_reward(self.h, coef.ctypes.data, len(coef), subtasks.ctypes.data, len(subtasks)))

    def set_attach_noise(self, noise, mask=None):
        """config.reset_robot_after_attach: the joint noise [n, narmjoints] the NEXT attach of each env adds to the arm's initial pose (fsim_set_attach_noise)"""
        nz = np.ascontiguousarray(noise, dtype=np.float32)
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.abi.check(self.abi.L.fsim_set_attach_noise(self.h, None if mk is None else mk.ctypes.data, nz.ctypes.data))

    def set_init_state(self, qpos, qvel, mask=None):
        """FurnitureEnv.set_init_qpos for the masked envs (qpos = None: back to sampled resets) -- fsim_set_init_state, host pointers on both libraries"""
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        if qpos is None:
            self.abi.check(self.abi.L.fsim_set_init_state(self.h, None if mk is None else mk.ctypes.data, None, None))
            return
        q, v = np.ascontiguousarray(qpos, dtype=np.float32), np.ascontiguousarray(qvel, dtype=np.float32)
        self.abi.check(self.abi.L.fsim_set_init_state(self.h, None if mk is None else mk.ctypes.data, q.ctypes.data, v.ctypes.data))

    def set_preassembled(self, model, preassembled, num_connects=None, welds=False):
        """config.preassembled / set_subtask / config.assembled (welds=True: the list holds weld ids whatever the furniture) -- fsim_set_preassembled, host
        pointers on both libraries; rows as furniture_amd.sim.FSim.set_preassembled builds them"""
        from furniture_amd.sim import preassembled_rows
        if welds:  # weld ids as they are: no recipe lookup
            ids, pairs, angles = np.ascontiguousarray(list(preassembled), dtype=np.int32), None, None
        else:
            ids, pairs, angles = preassembled_rows(model, preassembled)
        self.abi.check(self.abi.L.fsim_set_preassembled(self.h, len(ids), ids.ctypes.data if len(ids) else None, None if pairs is None else pairs.ctypes.data,
                                                        None if angles is None else angles.ctypes.data, -1 if num_connects is None else int(num_connects)))

    def reset(self):
        self.abi.check(self.abi.L.fsim_reset(self.h, None, self.abi.ptr(self.obs)))
        self.abi.check(self.abi.L.fsim_sync(self.h))
        return self.abi.get(self.obs)

    def step(self, actions):
        a = self.abi
        a.put(self.act, actions)
        a.check(a.L.fsim_step(self.h, a.ptr(self.act), a.ptr(self.obs), a.ptr(self.rew), a.ptr(self.done), a.ptr(self.info)))
        a.check(a.L.fsim_sync(self.h))
        return a.get(self.obs), a.get(self.rew), a.get(self.done), a.get(self.info)

    def forward(self):
        self.abi.check(self.abi.L.fsim_physics_forward(self.h))
        self.abi.check(self.abi.L.fsim_sync(self.h))

    def tables_needed(self):
        return self.abi.L.fsim_tables_needed(self.h)

    def _shape(self, name, m):
        n = self.n
        return {"qpos": ((n, self.nq), np.float32), "qvel": ((n, self.nv), np.float32), "qacc_warmstart": ((n, self.nv), np.float32),
                "qfrc_bias": ((n, self.nv), np.float32), "ctrl": ((n, self.nu), np.float32), "qfrc_applied": ((n, self.nv), np.float32),
                "xfrc_applied": ((n, m.nparts * 6), np.float32), "eq_data": ((n, m.neq * 7), np.float32), "eq_active": ((n, m.neq), np.int32),
                "geom_contype": ((n, m.ngeom), np.int32), "geom_conaffinity": ((n, m.ngeom), np.int32), "group": ((n, m.nparts), np.int32),
                "xpos": ((n, m.nbody * 3), np.float32), "xquat": ((n, m.nbody * 4), np.float32), "ncon": ((n,), np.int32),
                "contact_geoms": ((n, self.max_contacts * 2), np.int32), "cursor": ((n, 8), np.float32), "dense": ((n, 27), np.float32)}[name]

    def get_state(self, m, *names):
        a, sp, bufs = self.abi, StatePtrs(), {}
        for k in names:
            bufs[k] = a.zeros(*self._shape(k, m))
            setattr(sp, k, a.ptr(bufs[k]))
        a.check(a.L.fsim_get_state(self.h, ctypes.byref(sp)))
        a.check(a.L.fsim_sync(self.h))
        return {k: a.get(v) for k, v in bufs.items()}

    def set_state(self, m, **fields):
        a, sp, keep = self.abi, StatePtrs(), []
        for k, v in fields.items():
            shape, dt = self._shape(k, m)
            b = a.zeros(shape, dt)
            a.put(b, np.asarray(v).reshape(shape).astype(dt))
            keep.append(b)
            setattr(sp, k, a.ptr(b))
        a.check(a.L.fsim_set_state(self.h, ctypes.byref(sp)))
        a.check(a.L.fsim_sync(self.h))

    def close(self):
        if self.h:
            self.abi.L.fsim_destroy(self.h)
            self.h = None

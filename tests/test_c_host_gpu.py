"""The C-ABI without Python in the loop: examples/c_host.c (plain C, HIP runtime for the buffers, include/fsim.h for everything else) is
compiled with gcc, run as a process of its own, and every number it prints is reproduced by the same calls made through ctypes
(tests/abi_session.py) -- bit for bit.  What a reference-side cgo / JNI stub would do, exercised."""
import os
import struct
import subprocess

import numpy as np
import pytest

from furniture_amd.envs import ResetTableSampler, make_config
from tests.abi_session import Abi, Session, GPU_LIB, ROOT

pytestmark = pytest.mark.gpu


def _bits_sum(a):
    return int(np.ascontiguousarray(a).view(np.uint32).astype(np.uint64).sum())


def test_plain_c_host_reproduces_the_ctypes_session(sawyer_lack, tmp_path):
    import torch
    m, n, steps = sawyer_lack, 32, 8
    exe = str(tmp_path / "c_host")
    csrc = os.path.dirname(GPU_LIB)
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "examples", "c_host.c"), "-I" + os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", "-L" + csrc, "-L/opt/rocm/lib", "-lfsim", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + csrc, "-o", exe])
    blob = m.to_blob()
    (tmp_path / "model.blob").write_bytes(blob)
    ecfg = make_config(unity=False, record_vid=False, furniture_name="table_lack_0825", max_episode_steps=5, seed=3)
    parts, noise = ResetTableSampler(m, ecfg, 3, 0, n).draw()
    parts, noise = np.ascontiguousarray(parts, dtype=np.float32), np.ascontiguousarray(noise, dtype=np.float32)
    (tmp_path / "tables.bin").write_bytes(struct.pack("<ii", parts.shape[1], noise.shape[1]) + parts.tobytes() + noise.tobytes())
    out = subprocess.run([exe, str(tmp_path / "model.blob"), str(tmp_path / "tables.bin"), str(n), str(steps)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    # the same calls through ctypes
    ses = Session(Abi(GPU_LIB, torch.device("cuda:0")), blob, n, max_episode_steps=5, auto_reset=1)
    assert lines[0].startswith("kernel %s |" % ses.variant()) and ("dof %d obs %d" % (ses.dof, ses.obs_dim)) in lines[0]
    ses.set_reset_tables(parts, noise)
    obs = ses.reset()
    ses.set_reset_tables(parts, noise)
    assert lines[1] == "reset obs %d" % _bits_sum(obs)
    resets = 0
    for t in range(steps):
        e, k = np.meshgrid(np.arange(n), np.arange(ses.dof), indexing="ij")
        a = ((((e * 31 + k * 17 + t * 7) % 21) - 10).astype(np.float32) / np.float32(10.0)).astype(np.float32)
        obs, rew, done, info = ses.step(a)
        need = int((info[:, 7] != 0).sum())
        assert lines[2 + t] == "step %d obs %d reward %d done %d needs_table %d (fsim_tables_needed %d)" % (t, _bits_sum(obs), _bits_sum(rew), int(done.sum()), need, ses.tables_needed()), (t, lines[2 + t])
        if need:
            ses.set_reset_tables(parts, noise)
            resets += need
    assert resets == n  # the time limit at step 5 ended every env's episode once inside the run
    ses.close()

"""The CPU checker under AddressSanitizer + UndefinedBehaviorSanitizer: a reset, contact-rich random steps, a connect and a
Cursor episode run through `libfsim_oracle_san.so` in a subprocess (ASan must be the first library of the process: LD_PRELOAD).
The oracle is what every device test is compared against; an out-of-bounds read in it would be a silent error source."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import numpy as np
from furniture_amd.mjcf.model import load_compiled
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
from tests.scenarios import pinch_attach_state
m = load_compiled("Sawyer", "table_lack_0825")
env = FurnitureEnvOracle(m, OracleConfig(seed=5, max_episode_steps=30, preassembled=[0]))
env.reset()
rng = np.random.RandomState(0)
for t in range(6):
    ob, r, d, info = env.step(rng.uniform(-1, 1, 9))
assert np.isfinite(env.flat_obs(ob)).all()
c = load_compiled("Cursor", "toy_table")
cenv = FurnitureEnvOracle(c, OracleConfig(seed=2, max_episode_steps=30))
cenv.reset()
for t in range(4):
    cenv.step(rng.uniform(-1, 1, 15))
print("SANITIZED-OK")
"""


def test_oracle_runs_clean_under_asan_and_ubsan():
    so = os.path.join(ROOT, "oracle", "libfsim_oracle_san.so")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libfsim_oracle_san.so"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build not available: " + r.stderr[-200:])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, OSIM_LIB=so, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert p.returncode == 0 and "SANITIZED-OK" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-3000:]

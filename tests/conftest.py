import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sawyer_lack():
    from furniture_amd.mjcf.model import load_compiled
    return load_compiled("Sawyer", "table_lack_0825")


@pytest.fixture(scope="session")
def have_reference():
    from furniture_amd.mjcf.assemble import default_assets_root
    return default_assets_root() is not None


@pytest.fixture
def fsim_mw(request, monkeypatch):
    """Which step kernel the handles created inside the test run: parametrize indirectly with "0" (one wave per env), "1" (the
    scheduler's rule: k_env_step_x, multi-wave workgroups + bundles) or "all" (four waves for every env).  FSIM_MW is read by
    fsim_create and overrides fsim_config_t::multi_wave (include/fsim.h)."""
    mode = getattr(request, "param", None)
    if mode is not None:
        monkeypatch.setenv("FSIM_MW", mode)
    return mode


@pytest.fixture(scope="session", autouse=True)
def _poisoned_lds():
    """FSIM_TEST_POISON=<hex pattern> (e.g. 7fc00000): every fsim_step / fsim_reset / fsim_physics_* call of the session is preceded by a kernel that
    fills every CU's LDS with the pattern (tests/lds_poison.hip).  A test that passes without it and fails with it has found a code path that
    reads LDS before writing it.  Off by default (tests/test_lds_clean_gpu.py runs the comparison itself); the whole GPU suite is run this way
    by hand at the end of a round."""
    pat = os.environ.get("FSIM_TEST_POISON")
    if not pat:
        yield
        return
    import ctypes
    from furniture_amd import sim as S
    tool = ctypes.CDLL(os.path.join(ROOT, "tests", "liblds_poison.so"))
    saved = {}

    def wrap(name):
        orig = getattr(S.FSim, name)

        def f(self, *a, **k):
            for _ in range(2):
                tool.lds_poison(ctypes.c_uint(int(pat, 16)))
            return orig(self, *a, **k)
        saved[name] = orig
        setattr(S.FSim, name, f)
    for name in ("step", "reset", "physics_step", "physics_forward"):
        if hasattr(S.FSim, name):
            wrap(name)
    yield
    for name, orig in saved.items():
        setattr(S.FSim, name, orig)

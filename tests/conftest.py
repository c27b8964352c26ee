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

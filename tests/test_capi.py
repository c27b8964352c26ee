"""The C-ABI library loads and exports every entry point include/fsim.h declares; without a GPU it refuses loudly."""
import ctypes
import os
import re

import pytest
import torch

from furniture_amd import sim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "fsim.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fsim_[a-z_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(sim.build())
    names = _declared()
    assert set(names) == set(sim.EXPORTED_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_config_struct_matches_header_defaults():
    c = sim.default_config()
    assert (c.n_substeps, c.max_episode_steps, c.discrete_grip, c.rescale_actions, c.auto_align) == (50, 2000, 1, 1, 1)
    assert abs(c.alignment_pos_dist - 0.1) < 1e-7 and abs(c.pick_reward - 100) < 1e-7 and abs(c.agent_xyz_rand - 0.001) < 1e-9


def test_python_config_struct_has_the_headers_fields_in_the_headers_order():
    """fsim_config_t crosses the C-ABI by value: the ctypes mirror must list the header's fields, in its order, with its types -- a field
    added on one side only would shift everything behind it silently.  The LAST fields' defaults, read through fsim_default_config, pin
    the tail of the layout against the built library."""
    import re
    src = open(os.path.join(ROOT, "include", "fsim.h")).read()
    body = src[src.index("typedef struct fsim_config"):src.index("} fsim_config_t;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in re.findall(r"\b(int32_t|float)\s+([^;]+);", body):
        for name in decl[1].split(","):
            fields.append((name.strip(), decl[0]))
    py = [(n, {ctypes.c_int32: "int32_t", ctypes.c_float: "float"}[t]) for n, t in sim.FsimConfig._fields_]
    assert py == fields, [x for x in zip(py, fields) if x[0] != x[1]][:3]
    c = sim.default_config()
    assert (c.multi_wave, c.lookahead_reset, c.overflow_restep) == (0, 1, 1)


@pytest.mark.skipif(torch.cuda.is_available(), reason="exercises the no-GPU failure path")
def test_no_silent_cpu_fallback(sawyer_lack):
    # product classes refuse to construct without a device ...
    with pytest.raises(sim.FsimError):
        sim.FSim(sawyer_lack, 4)
    # ... and so does the raw C entry point (FSIM_ENODEV), with a message
    lib = sim.lib()
    blob = sawyer_lack.to_blob()
    h = ctypes.c_void_p()
    rc = lib.fsim_create(blob, len(blob), 4, 0, None, ctypes.byref(h))
    assert rc == -4 and b"no HIP device" in lib.fsim_last_error()
    assert lib.fsim_create(b"garbage" * 20, 140, 4, 0, None, ctypes.byref(h)) == -1


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "furniture_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "fsim_oracle" not in txt, f


def test_header_and_c_host_compile_as_plain_c(tmp_path):
    """include/fsim.h is a C header (no C++, no torch types): the plain-C host example compiles against it with gcc -std=c11 (object
    only here; the GPU suite links and runs it: tests/test_c_host_gpu.py)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no ROCm headers on this machine")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-c", os.path.join(root, "examples", "c_host.c"), "-I" + os.path.join(root, "include"),
                           "-I/opt/rocm/include", "-o", str(tmp_path / "c_host.o")])

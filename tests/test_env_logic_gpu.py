"""The DEVICE env-logic functions on the reference's golden vectors (through the C-ABI replay hooks of include/fsim.h).

tests/test_env_logic_golden.py pins the CPU restatement (oracle/oracle_env.py) to what the reference's own methods returned;
here the same vectors go straight to the code that runs inside fsim_step: env_is_aligned_core, env_connect_search +
env_connect_decide, fs_touch_flags + env_finger_scan (furniture_amd/csrc/fsim_env.hpp).  Integer results are exact; the
target quaternion is computed in fp32 from fp32-rounded inputs against the reference's fp64 (5e-6: the matrix -> quaternion
step divides by 2 sqrt(1 + trace), which amplifies the input rounding when the trace is near -1)."""
import os

import numpy as np
import pytest

from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, replay_is_aligned

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_device_is_aligned_on_reference_vectors():
    """FurnitureEnv._is_aligned (furniture.py:1057-1153): verdict and _target_connector_xquat of 400 recorded site-pose pairs."""
    G = np.load(os.path.join(GOLD, "env_logic.npz"))
    ok, tq = replay_is_aligned(G["p1"], G["R1"], G["p2"], G["R2"], G["nang"], G["angles"])
    ref_ok, ref_q = G["aligned"].astype(bool), G["target_quat"]
    # a verdict may only differ where an fp32 quantity sits on a threshold: none of the recorded cases does
    assert (ok == ref_ok).all(), np.nonzero(ok != ref_ok)[0]
    has = np.isfinite(ref_q).all(axis=1)
    assert (np.isfinite(tq).all(axis=1) == has).all()  # the quaternion is (not) written on exactly the reference's paths
    assert has.sum() > 100
    # q and -q are the same rotation, but the reference's lookat_to_quat branch is reproduced, so the signs agree too
    assert np.abs(tq[has] - ref_q[has]).max() < 5e-6, np.abs(tq[has] - ref_q[has]).max()
    assert np.median(np.abs(tq[has] - ref_q[has]).max(axis=1)) < 2e-7


def _union_find_parents(nparts, merges):
    grp = list(range(nparts))

    def find(i):  # furniture.py:2738-2753 (path compression changes the table, not the roots)
        if grp[i] == i:
            return i
        grp[i] = find(grp[i])
        return grp[i]

    for a, b in merges:
        if a >= 0:
            grp[find(int(a))] = find(int(b))
    return grp


def test_device_try_connect_on_reference_vectors(sawyer_lack):
    """FurnitureEnv._try_connect (furniture.py:926-1042) on the real connector / weld tables of Sawyer + table_lack_0825: which
    site pair is connected or approached, the return value, _connect_step and the part moved, for 120 recorded trials."""
    m = sawyer_lack
    G = np.load(os.path.join(GOLD, "env_logic.npz"))
    conn_site = [int(s) for s in m.conn_siteid]
    n, nc = len(G["tc_ret"]), len(conn_site)
    group = np.array([_union_find_parents(m.nparts, G["tc_merges"][t]) for t in range(n)], dtype=np.int32)
    used = G["tc_used"][:, conn_site].astype(np.int32)
    aligned = G["tc_aligned"][:, conn_site][:, :, conn_site].astype(np.uint8)
    part12 = np.stack([G["tc_part1"], G["tc_part2"]], axis=1).astype(np.int32)
    sim = FSim(m, 1)
    for nsteps in sorted(set(int(x) for x in G["tc_nsteps"])):
        sel = np.nonzero(G["tc_nsteps"] == nsteps)[0]
        out = sim.replay_try_connect(part12[sel], group[sel], used[sel], aligned[sel], G["tc_step_in"][sel], nsteps)
        for row, t in zip(out, sel):
            k1, k2, ret, step_out, moved = (int(x) for x in row)
            want_conn = tuple(int(x) for x in G["tc_conn"][t])
            got_conn = (conn_site[k1], conn_site[k2]) if ret else (-1, -1)
            assert bool(ret) == bool(G["tc_ret"][t]), t
            assert got_conn == want_conn, (t, got_conn, want_conn)
            assert step_out == int(G["tc_step_out"][t]), t
            assert moved == int(G["tc_moved"][t]), t
    sim.close()
    assert (G["tc_conn"][:, 0] >= 0).sum() >= 5 and (G["tc_moved"] >= 0).sum() >= 5


@pytest.mark.parametrize("agent,furn,narm", [("Sawyer", "table_lack_0825", 1), ("Baxter", "desk_mikael_1064", 2)])
def test_device_finger_touch_scan_on_reference_vectors(agent, furn, narm):
    """_step_continuous's connect scan (furniture.py:1290-1330) run by the reference on random contact lists: the parts
    _try_connect is asked about, in order.  The device holds colliding geoms only, so the trials whose lists name nothing else are
    replayed (58 / 67 of the 200 per agent); the touch masks are checked against the model tables as well."""
    m = load_compiled(agent, furn)
    S = np.load(os.path.join(GOLD, "step_scan.npz"))
    cg_of = {int(g): k for k, g in enumerate(m.cg_orig)}
    C, script, tried, connect = S[agent + "_contacts"], S[agent + "_script"], S[agent + "_tried"], S[agent + "_connect"]
    rows = [t for t in range(len(C)) if all(int(a) in cg_of and int(b) in cg_of for a, b in C[t] if a >= 0)]
    assert len(rows) >= 50
    maxc = C.shape[1]
    ncon = np.array([(C[t][:, 0] >= 0).sum() for t in rows], dtype=np.int32)
    geoms = np.zeros((len(rows), maxc, 2), dtype=np.int32)
    for r, t in enumerate(rows):
        for k in range(ncon[r]):
            geoms[r, k] = [cg_of[int(C[t][k, 0])], cg_of[int(C[t][k, 1])]]
    sim = FSim(m, 1)
    masks, got = sim.replay_touch_scan(ncon, geoms, script[rows])
    sim.close()
    nonempty = 0
    for r, t in enumerate(rows):
        # masks from first principles: bit arm*16 + part of L (R) <=> a left (right) finger geom of that arm touches a geom of that part
        wl = wr = 0
        for a, b in geoms[r, :ncon[r]]:
            for ga, gb in ((int(a), int(b)), (int(b), int(a))):
                part = int(m.cg_partid[gb])
                if part < 0:
                    continue
                role = int(m.cg_fingerrole[ga])
                for arm in range(narm):
                    if role & (1 << (2 * arm)):
                        wl |= 1 << (16 * arm + part)
                    if role & (1 << (2 * arm + 1)):
                        wr |= 1 << (16 * arm + part)
        assert (int(masks[r, 0]), int(masks[r, 1])) == (wl, wr), t
        want = [int(x) for x in tried[t] if x >= 0] if connect[t] > 0 else None
        if want is not None:  # (connect <= 0: the env does not scan at all, env_step tests connect > 0 before the scan)
            assert [int(x) for x in got[r] if x >= 0] == want, (agent, t, got[r], want)
            nonempty += bool(want)
    assert nonempty >= 10

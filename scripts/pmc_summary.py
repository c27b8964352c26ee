"""Per-dispatch PMC values of k_env_step from the rocprofv3 --pmc passes of scripts/profile_round.sh, and the derived
figures bench.py reports beside the HBM roofline (what actually binds the fused step kernel)."""
import glob
import json
import os
import sqlite3
import sys

N_SIMD, N_XCD = 1024, 8
ENVS = int(os.environ.get("FSIM_PMC_ENVS", "1024"))


def counters(db):
    c = sqlite3.connect(db)
    out = {}
    try:
        rows = c.execute("select counter_name, value, dispatch_id from counters_collection where kernel_name like '%k_env_step%' order by dispatch_id")
        for name, val, disp in rows:
            out.setdefault(name, {}).setdefault(disp, 0.0)
            out[name][disp] += val
    except Exception as e:  # schema differences between rocprofv3 versions
        print("pmc query failed for", db, e)
    return {k: [v[d] for d in sorted(v)] for k, v in out.items()}


def main(root, txt, js):
    allc = {}
    for db in sorted(glob.glob(root + "/pmc*/**/*.db", recursive=True)):
        allc.update(counters(db))
    lines = ["# rocprofv3 --kernel-trace --pmc passes (separate runs, see commands.txt), k_env_step / k_env_step_x, bench.py --steps 6 --warmup 1",
             "# (slabs of %d envs, Sawyer+table_lack_0825; PMC collection serialises the kernels); values per launch, summed over XCDs:" % ENVS,
             "# the slabs' reset launches first, then their warm-up + step launches in turn", ""]
    for k in sorted(allc):
        lines.append("%-28s " % k + " ".join("%.4g" % v for v in allc[k][-10:]))
    d = {}
    last = lambda k: allc[k][-1] if k in allc and allc[k] else None
    if last("SQ_INSTS_VALU"):
        valu, salu, lds = last("SQ_INSTS_VALU"), last("SQ_INSTS_SALU") or 0, last("SQ_INSTS_LDS") or 0
        smem, vrd, vwr = last("SQ_INSTS_SMEM") or 0, last("SQ_INSTS_VMEM_RD") or 0, last("SQ_INSTS_VMEM_WR") or 0
        wave_cyc, wait, active, stall = last("SQ_WAVE_CYCLES"), last("SQ_WAIT_ANY"), last("SQ_ACTIVE_INST_ANY"), last("SQ_WAIT_INST_ANY")
        gui = last("GRBM_GUI_ACTIVE")
        d.update(valu_insts_per_env_step=valu / ENVS, salu_insts_per_env_step=salu / ENVS, lds_insts_per_env_step=lds / ENVS,
                 smem_insts_per_env_step=smem / ENVS, vmem_insts_per_env_step=(vrd + vwr) / ENVS,
                 insts_per_env_step=(valu + salu + lds + smem + vrd + vwr) / ENVS)
        if wave_cyc:
            d.update(wait_frac=wait / wave_cyc, issue_frac=active / wave_cyc, issue_stall_frac=stall / wave_cyc,
                     wave_cycles_per_env_step=4.0 * wave_cyc / ENVS)  # SQ_* cycle counters tick in quad-cycles
        if gui:
            cyc = gui / N_XCD  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            d.update(kernel_cycles=cyc, valu_issue_util=valu / (N_SIMD * cyc),
                     waves_per_simd=4.0 * wave_cyc / (N_SIMD * cyc) if wave_cyc else None)
        mf = last("SQ_INSTS_VALU_MFMA_MOPS_F32")
        if mf is not None:
            d["mfma_f32_mops_per_launch"] = mf
    f, w = last("FETCH_SIZE"), last("WRITE_SIZE")
    if f is not None and w is not None:
        # rocprofv3 reports KB; record streams are dword-wide accesses (no 2x wide-read correction applies, MI355X_MICROARCH.md HBM)
        d.update(FETCH_SIZE_KB_per_launch=f, WRITE_SIZE_KB_per_launch=w, bytes_per_env_step=(f + w) * 1024.0 / ENVS)
    lines += ["", "derived (last step launch): " + json.dumps(d)]
    open(txt, "w").write("\n".join(lines) + "\n")
    json.dump(d, open(js, "w"), indent=1)
    print("\n".join(lines[-3:]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])

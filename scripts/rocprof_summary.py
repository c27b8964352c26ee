"""Summarise a rocprofv3 (--kernel-trace --stats) rocpd sqlite database into a small text file for profiles/."""
import sqlite3
import sys


def main(db, out, note=""):
    c = sqlite3.connect(db)
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s)" % db.split("/")[-1], note, "",
             "%-60s %8s %14s %14s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "pct")]
    for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 12"):
        lines.append("%-60s %8d %14.3f %14.3f %8.3f" % (name[:60], calls, tot / 1e3, avg / 1e3, pct))
    lines += ["", "# per-dispatch resources of the hot kernel (first dispatch)"]
    row = c.execute("select name,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count from kernels where name like '%k_env_step%' or name like '%k_physics%' limit 1").fetchone()
    if row:
        lines.append("kernel=%s grid=%d wg=%d lds_bytes=%d scratch_bytes_per_lane=%d vgpr=%d agpr=%d sgpr=%d" % ((row[0][:60],) + tuple(row[1:])))
    durs = [r[0] for r in c.execute("select duration from kernels where name like '%k_env_step%' order by start")]
    if durs:
        lines.append("k_env_step dispatch durations (ms): " + " ".join("%.1f" % (d / 1e6) for d in durs))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:8]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")

"""development: per-function instruction / scratch-instruction counts of a device assembly listing.  usage: funcs.py x.s [substring ...]"""
import re, subprocess, sys
want = sys.argv[2:] or ["fs_substeps_t", "fs_chol_mfma", "env_reset"]
name, n, sc, out = None, 0, 0, []
for line in open(sys.argv[1]):
    m = re.match(r"^(_ZL?\w+):", line)
    if m:
        name, n, sc = m.group(1), 0, 0
        continue
    if name and line.startswith(".Lfunc_end"):
        out.append((name, n, sc)); name = None
        continue
    if name and re.match(r"^\s+[a-z_0-9]+\s", line) and not line.lstrip().startswith("."):
        n += 1
        if "scratch_" in line: sc += 1
names = subprocess.run(["c++filt"], input="\n".join(o[0] for o in out), capture_output=True, text=True).stdout.split("\n")
for (mn, n, sc), dn in zip(out, names):
    if any(w in dn for w in want):
        print("%7d instr %5d scratch  %s" % (n, sc, dn[:170]))

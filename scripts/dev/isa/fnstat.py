"""development: per-function static instruction mix of a device assembly listing (see spills.sh for the flags).
usage: fnstat.py x.s substring [substring ...]  -- prints, for every function whose mangled name contains all substrings:
instructions, scratch loads / stores, v_readlane / v_writelane (SGPR spill traffic), v_accvgpr moves, and the scratch
clusters (runs of scratch instructions separated by > 200 lines: prologue / epilogue saves show up as the two big ones)."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2:]
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if m and all(w in m.group(1) for w in want):
        name = m.group(1)
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i:j]
        ins = [l for l in body if re.match(r"^\s[a-z]", l)]
        cnt = lambda p: sum(1 for l in ins if re.search(p, l))
        print("%s\n  instructions %d  scratch_load %d scratch_store %d  v_readlane %d v_writelane %d  accvgpr %d  s_waitcnt %d  ds_ %d  global/flat %d" % (
            name[:140], len(ins), cnt(r"scratch_load"), cnt(r"scratch_store"), cnt(r"v_readlane"), cnt(r"v_writelane"), cnt(r"v_accvgpr"),
            cnt(r"s_waitcnt"), cnt(r"^\sds_"), cnt(r"^\s(global|flat)_")))
        pos = [k for k, l in enumerate(body) if "scratch_" in l]
        if pos:
            cl, s, p, n = [], pos[0], pos[0], 1
            for k in pos[1:]:
                if k - p > 200:
                    cl.append((s, p, n)); s, n = k, 0
                p = k; n += 1
            cl.append((s, p, n))
            print("  scratch clusters (line range, count):", " ".join("%d-%d(%d)" % c for c in cl), " of", len(body), "lines")
        i = j
    i += 1

"""development: static instruction counts per kernel of an AMDGPU .s file, by class"""
import re, sys, collections
cur = None
cnt = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"^(\w+):\s*(;.*)?$", line)
    if m and not m.group(1).startswith(("BB", "Lfunc", "LBB")):
        cur = m.group(1); cnt[cur] = collections.Counter(); continue
    if line.startswith(".Lfunc_end"): cur = None
    if cur is None: continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", line)
    if not m: continue
    op = m.group(1)
    c = cnt[cur]
    c["all"] += 1
    if op.startswith("ds_"): c["lds"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): c["vmem"] += 1; c["scratch"] += op.startswith("scratch_")
    elif op.startswith("s_waitcnt"): c["wait"] += 1
    elif op.startswith("s_nop"): c["nop"] += 1
    elif op.startswith("s_load"): c["sload"] += 1
    elif op.startswith("s_"): c["salu"] += 1
    elif op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): c["lane"] += 1
    elif op.startswith("v_mov") or op.startswith("v_accvgpr"): c["vmov"] += 1
    elif op.startswith("v_cndmask"): c["cnd"] += 1
    elif op.startswith("v_cmp"): c["cmp"] += 1
    elif op.startswith("v_"): c["valu"] += 1
print("%-24s %6s %5s %5s %5s %5s %5s %5s %5s %5s %5s %5s %5s" % ("kernel", "all", "valu", "vmov", "cnd", "cmp", "lane", "salu", "lds", "vmem", "scr", "wait", "nop"))
for k, c in cnt.items():
    if c["all"] < 5: continue
    print("%-24s %6d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d" % (k[:24], c["all"], c["valu"], c["vmov"], c["cnd"], c["cmp"], c["lane"], c["salu"], c["lds"], c["vmem"], c["scratch"], c["wait"], c["nop"]))

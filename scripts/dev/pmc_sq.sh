#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/pmc_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o pmc -- python $R/bench.py --steps 6 --warmup 1 --groups 1 --no-cpu-baseline > $O/p$i.log 2>&1
  DBP=$(find $O/p$i -name "*.db" | head -1)
  [ -n "$DBP" ] && python - "$DBP" <<'PY'
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select counter_name, dispatch_id, value from counters_collection where kernel_name like 'k_env_step%' order by dispatch_id"))
acc = collections.OrderedDict()
for name, did, v in rows:
    acc.setdefault(name, collections.OrderedDict()).setdefault(did, 0.0)
    acc[name][did] += v
for name, d in acc.items():
    print("%-22s %s" % (name, " ".join("%.3e" % v for v in d.values())))
PY
done

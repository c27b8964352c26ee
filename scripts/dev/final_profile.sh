#!/bin/bash
# end-of-round measurement: default bench line, rocprofv3 kernel-trace stats of the same command, HBM-traffic PMC passes
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 30 --warmup 2 --groups 1 --no-cpu-baseline > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
[ -n "$DB" ] && python $R/scripts/rocprof_summary.py $DB $O/kt_summary.txt "bench.py --steps 30 --warmup 2 --groups 1 (single stream), 4096 envs Sawyer+table_lack_0825, round 1 final (after config 5 / f2 / f3 / plane-cylinder conditioning)" | tail -8
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt2 -o kt2 -- python $R/bench.py --steps 30 --warmup 2 --no-cpu-baseline > $O/kt2.log 2>&1
DB2=$(find $O/kt2 -name "*.db" | head -1)
[ -n "$DB2" ] && python $R/scripts/rocprof_summary.py $DB2 $O/kt2_summary.txt "bench.py --steps 30 --warmup 2 (default: 2 slabs of 2048 on separate streams), round 1 final" | tail -4
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 5 --warmup 1 --groups 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1
  DBP=$(find $O/pmc_$c -name "*.db" | head -1)
  [ -n "$DBP" ] && python - "$DBP" $c <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
try:
    rows = list(c.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like 'k_env_step%' order by dispatch_id"))
    vals = [r[2] for r in rows]
    print(sys.argv[2], "per k_env_step dispatch:", " ".join("%.4g" % v for v in vals))
except Exception as e:
    print("pmc query failed:", e, [t for t in tabs if 'counter' in t.lower() or 'pmc' in t.lower()])
PY
done
ls $O

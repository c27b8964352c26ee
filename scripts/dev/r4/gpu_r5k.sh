#!/bin/bash
# k_schedule: keys in flight per lane (8 / 4 / 2): duration under rocprofv3 and bench
R=$PWD; O=$R/gpurun_out/r5k; mkdir -p $O
export PYTHONPATH=$R
C=$R/furniture_amd/csrc
cd /tmp && export TMPDIR=/tmp
for l in libfsim libfsim_u4 libfsim_u2; do
  FSIM_LIB=$C/$l.so timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$l -o kt -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/kt_$l.log 2>&1
  DB=$(find $O/kt_$l -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/scripts/rocprof_summary.py $DB $O/kt_$l.txt "$l" | grep "k_schedule\|k_env_step_x"
  rm -rf $O/kt_$l
done
cd $R
for k in 1 2; do for l in libfsim libfsim_u4 libfsim_u2; do
  FSIM_LIB=$C/$l.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${l}_${k}_20_5.json 2> $O/err.txt
  FSIM_LIB=$C/$l.so python bench.py --no-lookahead --steps 100 --warmup 10 --no-cpu-baseline > $O/${l}_${k}_100_10.json 2> $O/err.txt
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5k/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3))
    except Exception as e: print(f, "ERR", e)
PY

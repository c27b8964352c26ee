#!/bin/bash
# kernel-entry spills of the persistent launch: grid n/4 + n/16 (libfsim.so) and the opaque thread index on top (libfsim_op.so)
R=$PWD; O=$R/gpurun_out/r5b; mkdir -p $O
export PYTHONPATH=$R
C=$R/furniture_amd/csrc
for k in 1 2 3; do
  for l in libfsim libfsim_op; do
    FSIM_LIB=$C/$l.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${l}_${k}_20_5.json 2> $O/err.txt
    FSIM_LIB=$C/$l.so python bench.py --no-lookahead --steps 100 --warmup 10 --no-cpu-baseline > $O/${l}_${k}_100_10.json 2> $O/err.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for l in libfsim libfsim_op; do
  for set in WRITE_SIZE FETCH_SIZE; do
    FSIM_LIB=$C/$l.so timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_${l}_$set -o pmc -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline > $O/pmc_${l}_$set.log 2>&1
  done
done
cd $R
python - <<'PY'
import json,glob,sqlite3
for f in sorted(glob.glob("gpurun_out/r5b/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3))
    except Exception as e: print(f, "ERR", e)
for l in ("libfsim","libfsim_op"):
    for cset in ("WRITE_SIZE","FETCH_SIZE"):
        dbs=glob.glob("gpurun_out/r5b/pmc_%s_%s/**/*.db"%(l,cset),recursive=True)
        if not dbs: print(l,cset,"no db"); continue
        c=sqlite3.connect(dbs[0])
        try:
            rows=c.execute("select dispatch_id, sum(value) from counters_collection where kernel_name like '%k_env_step_x%' group by dispatch_id order by dispatch_id").fetchall()
            v=[r[1] for r in rows][-8:]
            print(l,cset,"KB per launch (last 8):",[round(x) for x in v]," per env-step KB: %.1f"%(sum(v)/len(v)/1024))
        except Exception as e: print(l,cset,"query failed",e)
PY
rm -rf $O/pmc_*_SIZE

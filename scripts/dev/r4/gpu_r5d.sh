#!/bin/bash
# the other BASELINE configurations with the final library
O=gpurun_out/r5d; mkdir -p $O
export PYTHONPATH=$PWD
for c in 3 4 5; do
  timeout 600 python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline > $O/config${c}_100_10.json 2> $O/config${c}.err
done
timeout 600 python bench.py --dense --steps 100 --warmup 10 --no-cpu-baseline > $O/dense_100_10.json 2> $O/dense.err
timeout 600 python bench.py --control-type position_orientation --steps 100 --warmup 10 --no-cpu-baseline > $O/osc_100_10.json 2> $O/osc.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5d/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), d["config"].get("workload","")[:80], d["roofline"].get("kernel"))
    except Exception as e: print(f, "ERR", e, open(f.replace("_100_10.json",".err")).read()[-300:])
PY

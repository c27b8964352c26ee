#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O
export PYTHONPATH=$PWD
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/c2.json 2> $O/c2.err
timeout 600 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline > $O/c5.json 2> $O/c5.err
python - <<'PY'
import json
for f in ("c2","c5"):
    try: d=json.loads(open("gpurun_out/r5f/%s.json"%f).read().strip().splitlines()[-1]); print(f, round(d["value"]), d["config"].get("envs_that_dropped_contacts"), d["config"].get("contact_slots"))
    except Exception as e: print(f,"ERR",e, open("gpurun_out/r5f/%s.err"%f).read()[-400:])
PY

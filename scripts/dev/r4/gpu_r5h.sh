#!/bin/bash
O=gpurun_out/r5h; mkdir -p $O
export PYTHONPATH=$PWD
for k in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/on${k}_20_5.json 2> $O/err.txt
  FSIM_NO_OVERFLOW_REDO=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/off${k}_20_5.json 2> $O/err.txt
  python bench.py --no-lookahead --steps 100 --warmup 10 --no-cpu-baseline > $O/on${k}_100_10.json 2> $O/err.txt
  FSIM_NO_OVERFLOW_REDO=1 python bench.py --no-lookahead --steps 100 --warmup 10 --no-cpu-baseline > $O/off${k}_100_10.json 2> $O/err.txt
done
python bench.py --no-cpu-baseline > $O/on_default.json 2> $O/err.txt
FSIM_NO_OVERFLOW_REDO=1 python bench.py --no-cpu-baseline > $O/off_default.json 2> $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5h/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), "dropped", d["config"]["envs_that_dropped_contacts"], "resteps", d["config"]["overflow_resteps"])
    except Exception as e: print(f, "ERR", e)
PY

#!/bin/bash
O=gpurun_out/r4o2; mkdir -p $O
export PYTHONPATH=$PWD
python bench.py --no-lookahead --groups 1 --multi-wave rule --steps 20 --warmup 5 > $O/g1r_20_5.json 2> $O/g1r_20_5.err
python bench.py --no-lookahead --groups 1 --multi-wave rule --steps 100 --warmup 10 > $O/g1r_100_10.json 2> $O/g1r_100_10.err
python bench.py --groups 1 --multi-wave rule > $O/g1r_default.json 2> $O/g1r_default.err
python bench.py --groups 2 > $O/g2_default.json 2> $O/g2_default.err
python bench.py --groups 4 > $O/g4_default.json 2> $O/g4_default.err
python bench.py --groups 2 > $O/g2_default_b.json 2> $O/g2_default.err
python bench.py --groups 4 > $O/g4_default_b.json 2> $O/g4_default.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4o2/*.json")):
    try: d=json.load(open(f)); print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d["roofline"].get("kernel"))
    except Exception as e: print(f, "ERR", e)
PY

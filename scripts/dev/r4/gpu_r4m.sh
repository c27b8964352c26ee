#!/bin/bash
# round 4 call M: look-ahead policy sweep on the default protocol (1000-step episodes, 100 + 1000 steps)
O=gpurun_out/r4m; mkdir -p $O
export PYTHONPATH=$PWD
r() { local name=$1; shift; env "$@" python bench.py > $O/$name.json 2> $O/$name.err; }
r default
r defer0 FSIM_LA_DEFER=0
r defer500 FSIM_LA_DEFER=500
r defer800 FSIM_LA_DEFER=800
r chunk26 FSIM_LA_CHUNK=26
r chunk101 FSIM_LA_CHUNK=101
r jobs8 FSIM_LA_JOBS=8
r jobs32 FSIM_LA_JOBS=32
r nola FSIM_NO_LOOKAHEAD=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4m/*.json")):
    try: d=json.load(open(f)); print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d.get("lookahead"))
    except Exception as e: print(f, "ERR", e)
PY

#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
export PYTHONPATH=$PWD
C=$PWD/furniture_amd/csrc
b() { local name=$1 lib=$2; shift 2
  env FSIM_LIB=$C/$lib "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${name}_20_5.json 2> $O/${name}_20_5.err
  env FSIM_LIB=$C/$lib "$@" python bench.py --no-lookahead --steps 100 --warmup 10 --no-cpu-baseline > $O/${name}_100_10.json 2> $O/${name}_100_10.err
}
b prev libfsim_prev.so
b lean libfsim_lean.so
b prev2 libfsim_prev.so
b lean2 libfsim_lean.so
b prev3 libfsim_prev.so
b lean3 libfsim_lean.so
env FSIM_LIB=$C/libfsim_lean.so python bench.py --no-cpu-baseline > $O/lean_default.json 2> $O/lean_default.err
env FSIM_LIB=$C/libfsim_prev.so python bench.py --no-cpu-baseline > $O/prev_default.json 2> $O/prev_default.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4s/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"]), d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY

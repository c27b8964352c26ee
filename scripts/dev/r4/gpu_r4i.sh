#!/bin/bash
# round 4 call I: scheduler reads the iteration rate of the last ten substeps -- A/B against the whole-step count on one box
O=gpurun_out/r4i; mkdir -p $O
export PYTHONPATH=$PWD
C=$PWD/furniture_amd/csrc
b() { # name lib extra-env...
  local name=$1 lib=$2; shift 2
  env FSIM_LIB=$C/$lib "$@" python bench.py --no-lookahead --steps 20 --warmup 5 > $O/${name}_20_5.json 2> $O/${name}_20_5.err
  env FSIM_LIB=$C/$lib "$@" python bench.py --no-lookahead --steps 100 --warmup 10 > $O/${name}_100_10.json 2> $O/${name}_100_10.err
}
b tail libfsim.so
b notail libfsim_notail.so
b tail_k120 libfsim.so FSIM_MW_K=120
b tail_k200 libfsim.so FSIM_MW_K=200
b tail_k250 libfsim.so FSIM_MW_K=250
b tail2 libfsim.so
b notail2 libfsim_notail.so
FSIM_LIB=$C/libfsim.so python bench.py > $O/tail_default.json 2> $O/tail_default.err
FSIM_TL_DUMP=$O/tl_rule.npy FSIM_LIB=$C/libfsim_tl.so timeout 300 python scripts/dev/timeline_x.py 50 rule > $O/timeline_rule.txt 2>&1
for t in test_determinism_gpu test_lookahead_gpu test_gpu_parity; do timeout 600 python -m pytest tests/$t.py -x -q 2>&1 | tail -4 > $O/$t.txt; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4i/*.json")):
    try: d=json.load(open(f)); print(f.split("/")[-1], round(d["value"]), d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
tail -2 $O/test_*.txt

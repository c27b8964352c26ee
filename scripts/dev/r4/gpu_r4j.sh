#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
export PYTHONPATH=$PWD
C=$PWD/furniture_amd/csrc
timeout 300 python scripts/dev/r4_diag.py grevsub > $O/grevsub.txt 2>&1
b() { local name=$1 lib=$2; shift 2
  env FSIM_LIB=$C/$lib "$@" python bench.py --no-lookahead --steps 20 --warmup 5 > $O/${name}_20_5.json 2> $O/${name}_20_5.err
  env FSIM_LIB=$C/$lib "$@" python bench.py --no-lookahead --steps 100 --warmup 10 > $O/${name}_100_10.json 2> $O/${name}_100_10.err
}
b base libfsim.so
b mfma4w libfsim_mfma4w.so
b base2 libfsim.so
b mfma4w2 libfsim_mfma4w.so
b mfma4w_k120 libfsim_mfma4w.so FSIM_MW_K=120
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4j/*.json")):
    try: d=json.load(open(f)); print(f.split("/")[-1], round(d["value"]), d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
tail -8 $O/grevsub.txt

#!/bin/bash
# matrix-core Hessian assembly only for the islands that are on the MFMA tile anyway (> 16 dofs): -DFSIM_MFMA_HESSIAN -DFSIM_BIG_MIN=17
O=gpurun_out/r5m; mkdir -p $O
export PYTHONPATH=$PWD
C=$PWD/furniture_amd/csrc
for k in 1 2 3; do for l in libfsim libfsim_mfma17; do
  FSIM_LIB=$C/$l.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${l}_${k}_20_5.json 2> $O/err.txt
  FSIM_LIB=$C/$l.so python bench.py --no-lookahead --steps 100 --warmup 10 --no-cpu-baseline > $O/${l}_${k}_100_10.json 2> $O/err.txt
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5m/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3))
    except Exception as e: print(f, "ERR", e)
PY

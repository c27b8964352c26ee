#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
export PYTHONPATH=$PWD
export FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_spin.so
for k in 1 2 3; do
  FSIM_SYNC_SPIN=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/sleep${k}_20_5.json 2> $O/err.txt
  FSIM_SYNC_SPIN=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/spin${k}_20_5.json 2> $O/err.txt
  FSIM_SYNC_SPIN=0 python bench.py --no-lookahead --steps 100 --warmup 10 --no-cpu-baseline > $O/sleep${k}_100_10.json 2> $O/err.txt
  FSIM_SYNC_SPIN=1 python bench.py --no-lookahead --steps 100 --warmup 10 --no-cpu-baseline > $O/spin${k}_100_10.json 2> $O/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4z/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), round(d["roofline"]["kernel_avg_ms"],3))
    except Exception as e: print(f, "ERR", e)
PY

#!/bin/bash
# workgroups per k_env_step_x launch (four slabs of 1024 envs): does a smaller grid per slab pack the chip better?
O=gpurun_out/r4v; mkdir -p $O
export PYTHONPATH=$PWD
export FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_grid.so
for g in 384 128 160 192 256 320 512; do
  FSIM_X_GRID=$g python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/grid${g}_20_5.json 2> $O/grid${g}_20_5.err
  FSIM_X_GRID=$g python bench.py --no-lookahead --steps 100 --warmup 10 --no-cpu-baseline > $O/grid${g}_100_10.json 2> $O/grid${g}_100_10.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4v/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), d["roofline"]["kernel_avg_ms"])
    except Exception as e: print(f, "ERR", e)
PY

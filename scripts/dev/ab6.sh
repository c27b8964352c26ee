#!/bin/bash
# workload-realisation noise: same lib, different env / action seeds
for v in "$@"; do
for seed in 123 1123 2123 3123; do
  lib=furniture_amd/csrc/libfsim_$v.so
  FSIM_BENCH_SEED=$seed FSIM_LIB=$PWD/$lib timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v seed=$seed value=%.0f ms/step=%.2f kernel_avg_ms=%.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))
"
done
done

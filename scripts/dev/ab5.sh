#!/bin/bash
# same-box A/B of lib variants, interleaved repetitions
for rep in 1 2 3; do
for v in "$@"; do
  lib=furniture_amd/csrc/libfsim_$v.so
  FSIM_LIB=$PWD/$lib timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep$rep $v value=%.0f ms/step=%.2f kernel_avg_ms=%.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))
"
done
done

#!/bin/bash
# ab.sh VARIANT... : bench.py (no cpu baseline) once per lib variant, same box, prints value + kernel avg
for v in "$@"; do
  lib=furniture_amd/csrc/libfsim.so; [ "$v" != base ] && lib=furniture_amd/csrc/libfsim_$v.so
  for g in 1 2; do
    FSIM_LIB=$PWD/$lib timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 2 --groups $g 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v groups=$g value=%.0f ms/step=%.2f kernel_avg_ms=%.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))
"
  done
done

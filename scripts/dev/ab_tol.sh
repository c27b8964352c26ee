#!/bin/bash
for tol in 1e-6 3e-6 1e-5 3e-5 1e-4; do
  FSIM_BENCH_TOL=$tol timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tol=$tol value=%.0f ms/step=%.2f kernel_avg_ms=%.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))
"
done

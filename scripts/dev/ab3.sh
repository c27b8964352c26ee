#!/bin/bash
for g in 2 4 8 16; do
  timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 2 --groups $g 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('in-order groups=$g value=%.0f ms/step=%.2f' % (d['value'], d['ms_per_step']))
"
  FSIM_BENCH_FREE_RUN=1 timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 2 --groups $g 2>&1 | grep -E "free_run|rror" | cut -c1-200
done

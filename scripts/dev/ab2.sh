#!/bin/bash
# default bench (both group settings) + controller exploration runs
for g in 1 2; do
  timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 2 --groups $g 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('default groups=$g value=%.0f ms/step=%.2f kernel_avg_ms=%.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))
"
done
for ct in "$@"; do
  timeout 120 python bench.py --no-cpu-baseline --steps 60 --warmup 2 --control-type $ct 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$ct value=%.0f ms/step=%.2f kernel_avg_ms=%.2f finite=%s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['config']['obs_finite']))
    elif 'rror' in l: print(l.strip()[:200])
"
done

#!/bin/bash
# round 6 (third session): a minimum distance between the launches of two slabs (host side only: FSIM_BENCH_SPACING_MS)
R=$PWD; O=$R/gpurun_out/r6sp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$2: %.0f env-steps/s  %.3f ms/step  kernel %.3f ms x %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_launches']))
"; }
for sp in ${SP_LIST:-0 0.3 0.6 0.9 1.2 0}; do
  for rep in 1 2; do
    FSIM_BENCH_SPACING_MS=$sp timeout 120 python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/w200_s${sp}_$rep.json 2> $O/w200_s${sp}_$rep.err || echo "rc $?"
    line $O/w200_s${sp}_$rep.json "spacing $sp ms, 200 steps"
  done
done

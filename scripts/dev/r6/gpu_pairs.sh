#!/bin/bash
# round 6 (third session): paired slab launches (host side only: FSIM_BENCH_PAIRS=1, FSIM_BENCH_PAIR_SPACING_MS)
R=$PWD; O=$R/gpurun_out/r6pairs; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$2: %.0f env-steps/s  %.3f ms/step  kernel %.3f ms x %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_launches']))
"; }
for sp in ${SP_LIST:-off 0 1.5 2.0 2.4 off 2.0}; do
  for rep in 1 2; do
    if [ $sp = off ]; then export FSIM_BENCH_PAIRS=0; else export FSIM_BENCH_PAIRS=1 FSIM_BENCH_PAIR_SPACING_MS=$sp; fi
    timeout 120 python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/w200_${sp}_$rep.json 2> $O/w200_${sp}_$rep.err || echo "rc $?"
    line $O/w200_${sp}_$rep.json "pairs $sp, 200 steps"
  done
done

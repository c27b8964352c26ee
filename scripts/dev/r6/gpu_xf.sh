#!/bin/bash
# round 6 (third session): the first FSIM_X_FIRST workgroups of a k_env_step_x launch go straight to the bundle queue (the longest one-wave jobs start
# with the first teams instead of behind them); timeline of who ends a launch, then A/B
R=$PWD; O=$R/gpurun_out/r6xf; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=$R/scripts/dev/r6/solo_vs_four.py
for xf in 0 8; do
  export OUT=/tmp/svf_$xf FSIM_X_FIRST=$xf
  LAYOUT=solo timeout 200 python $S 40 > /dev/null 2>&1; LAYOUT=shared timeout 200 python $S 40 > /dev/null 2>&1; timeout 100 python $S 40 > $O/timeline_xfirst_$xf.txt 2>&1
  echo "== FSIM_X_FIRST=$xf"; tail -3 $O/timeline_xfirst_$xf.txt | cut -c1-600
done
unset OUT
export FSIM_LIB=$R/furniture_amd/csrc/libfsim_xf.so
line() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$2: %.0f env-steps/s  %.3f ms/step  kernel %.3f ms x %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_launches']))
"; }
for xf in 0 4 8 16 0 8; do
  for rep in 1 2; do
    FSIM_X_FIRST=$xf timeout 120 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/w100_x${xf}_$rep.json 2> $O/w100_x${xf}_$rep.err || echo "rc $?"
    line $O/w100_x${xf}_$rep.json "bundle-first workgroups $xf, 100 steps"
  done
done

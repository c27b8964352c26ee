#!/bin/bash
# round 6 (third session): the first workgroups of a slab's step on a HIGH-PRIORITY stream of the handle (no CUs set aside): FSIM_LONG_PRIO = their number.
R=$PWD; O=$R/gpurun_out/r6x3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export FSIM_LIB=${FSIM_LIB:-$R/furniture_amd/csrc/libfsim_rsv.so}
line() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$2: %.0f env-steps/s  %.3f ms/step  kernel %.3f ms x %d  finite %s' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_launches'], d['config'].get('obs_finite')))
"; }
for n in ${PRIO_LIST:-0 16 32 64 0 32}; do
  for rep in 1 2; do
    FSIM_LONG_PRIO=$n timeout 120 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/w100_p${n}_$rep.json 2> $O/w100_p${n}_$rep.err || echo "rc $? (prio $n)"
    line $O/w100_p${n}_$rep.json "high-priority workgroups $n, 100 steps"
  done
  FSIM_LONG_PRIO=$n timeout 120 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_p${n}.json 2> $O/drv_p${n}.err || echo "rc $? (prio $n)"
  line $O/drv_p${n}.json "high-priority workgroups $n, driver command"
done

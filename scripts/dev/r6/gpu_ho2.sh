#!/bin/bash
# hand-over variant: parity tests first (generic kernels too), then the bench over the rule's threshold
R=$PWD; O=$R/gpurun_out/r6ho2; mkdir -p $O
export FSIM_LIB=$R/furniture_amd/csrc/libfsim_ho.so
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_determinism_gpu.py tests/test_deferred_resets_gpu.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$2: %.0f env-steps/s  %.3f ms/step  kernel %.3f ms x %d  finite %s' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_launches'], d['config'].get('obs_finite')))
"; }
export FSIM_HANDOVER_STATS=1
for it in ${IT_LIST:-off 36 30 24 18 off 30}; do
  for rep in 1 2; do
    if [ $it = off ]; then export FSIM_NO_HANDOVER=1; else unset FSIM_NO_HANDOVER; export FSIM_HANDOVER_ITERS=$it; fi
    timeout 120 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/w100_${it}_$rep.json 2> $O/w100_${it}_$rep.err || echo "rc $?"
    line $O/w100_${it}_$rep.json "threshold $it, 100 steps"; grep 'handed over' $O/w100_${it}_$rep.err | head -1 | cut -c20-120
  done
done

#!/bin/bash
# round 6 (third session): the hand-over in mid-step (libfsim_ho.so) against the shipped library; FSIM_NO_HANDOVER=1 on the same library as the control
R=$PWD; O=$R/gpurun_out/r6ho; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LIBHO=$R/furniture_amd/csrc/libfsim_${HO_TAG:-ho}.so
line() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$2: %.0f env-steps/s  %.3f ms/step  kernel %.3f ms x %d  finite %s' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_launches'], d['config'].get('obs_finite')))
"; }
export FSIM_HANDOVER_STATS=1
for rep in 1 2; do
  FSIM_LIB=$R/furniture_amd/csrc/libfsim.so timeout 120 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/w100_base_$rep.json 2> $O/w100_base_$rep.err || echo "rc $?"
  line $O/w100_base_$rep.json "shipped, 100 steps"
  FSIM_LIB=$LIBHO FSIM_NO_HANDOVER=1 timeout 120 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/w100_off_$rep.json 2> $O/w100_off_$rep.err || echo "rc $?"
  line $O/w100_off_$rep.json "variant, hand-over off, 100 steps"
  FSIM_LIB=$LIBHO timeout 120 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/w100_on_$rep.json 2> $O/w100_on_$rep.err || echo "rc $?"
  line $O/w100_on_$rep.json "variant, hand-over on, 100 steps"; grep 'handed over' $O/w100_on_$rep.err | head -4
done
for tag in base on; do
  lib=$LIBHO; [ $tag = base ] && lib=$R/furniture_amd/csrc/libfsim.so
  FSIM_LIB=$lib timeout 120 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_${tag}.json 2> $O/drv_${tag}.err || echo "rc $?"
  line $O/drv_${tag}.json "$tag, driver command"
done
if [ -n "$1" ]; then
  cd $R && FSIM_LIB=$LIBHO timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_determinism_gpu.py tests/test_capi_cpu.py tests/test_baseline_configs_gpu.py tests/test_deferred_resets_gpu.py tests/test_overflow_restep_gpu.py -m gpu -x -q 2>&1 | tail -8
fi

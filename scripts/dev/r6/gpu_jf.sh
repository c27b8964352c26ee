#!/bin/bash
# round 6 (third session): A/B of a library variant against the shipped one on one box: gpu_jf.sh <tag> [tests]
#   100-step window x 3 and the driver's command, alternating; then (tests) the parity / determinism files of the GPU suite on the variant
R=$PWD; TAG=${1:-jf}; O=$R/gpurun_out/r6$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
line() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$2: %.0f env-steps/s  %.3f ms/step  kernel %.3f ms x %d  finite %s' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_launches'], d['config'].get('obs_finite')))
"; }
for rep in 1 2 3; do
  for tag in base $TAG; do
    lib=$R/furniture_amd/csrc/libfsim_$tag.so; [ $tag = base ] && lib=$R/furniture_amd/csrc/libfsim.so
    FSIM_LIB=$lib timeout 120 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/w100_${tag}_$rep.json 2> $O/w100_${tag}_$rep.err || echo "rc $?"
    line $O/w100_${tag}_$rep.json "$tag, 100 steps"
  done
done
for tag in base $TAG base $TAG; do
  lib=$R/furniture_amd/csrc/libfsim_$tag.so; [ $tag = base ] && lib=$R/furniture_amd/csrc/libfsim.so
  FSIM_LIB=$lib timeout 120 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_${tag}.json 2> $O/drv_${tag}.err || echo "rc $?"
  line $O/drv_${tag}.json "$tag, driver command"
done
if [ -n "$2" ]; then
  cd $R && FSIM_LIB=$R/furniture_amd/csrc/libfsim_$TAG.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_determinism_gpu.py tests/test_capi_cpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q 2>&1 | tail -5
fi

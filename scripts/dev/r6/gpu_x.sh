#!/bin/bash
# round 6 (third session): reserved CUs for a slab's long jobs (FSIM_RESERVE = CUs per XCD per slab; the step = two launches of k_env_step_x on
# the same queues, the first on the handle's own CUs).  A/B on one box, same library: off / 1 / 2 / 3, 100-step window twice + the driver's command.
R=$PWD; O=$R/gpurun_out/r6x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export FSIM_LIB=${FSIM_LIB:-$R/furniture_amd/csrc/libfsim_rsv.so}
line() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$2: %.0f env-steps/s  %.3f ms/step  kernel %.3f ms x %d  finite %s' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_launches'], d['config'].get('obs_finite')))
"; }
for rsv in ${RSV_LIST:-0 1 2 3 0 2}; do
  for rep in 1 2; do
    FSIM_RESERVE=$rsv timeout 120 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/w100_r${rsv}_$rep.json 2> $O/w100_r${rsv}_$rep.err || echo "rc $? (reserve $rsv)"
    line $O/w100_r${rsv}_$rep.json "reserve $rsv, 100 steps"
  done
  FSIM_RESERVE=$rsv timeout 120 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_r${rsv}.json 2> $O/drv_r${rsv}.err || echo "rc $? (reserve $rsv)"
  line $O/drv_r${rsv}.json "reserve $rsv, driver command"
done

#!/bin/bash
# round 6: parameter sweeps on the committed library (threads default): multi-wave threshold K, slabs per GPU, hardware queues
O=gpurun_out/r6d; mkdir -p $O
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0"
run() { tag=$1; shift; "$@" > $O/$tag.json 2> $O/$tag.err; echo $tag $(python -c "import json; d=json.load(open('$O/$tag.json')); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_avg_ms'],3))" 2>&1 | tail -1); }
for rep in 1 2; do
  run base_$rep $B
  for k in 100 120 135 175 200; do FSIM_MW_K=$k run k${k}_$rep $B; done
  run groups8_$rep env GPU_MAX_HW_QUEUES=8 $B --groups 8
  run groups8q4_$rep $B --groups 8
  run groups2_$rep $B --groups 2
  run groups6_$rep env GPU_MAX_HW_QUEUES=8 $B --envs-per-gpu 4092 --groups 6
  run threads0_$rep $B --threads 0
done
run default_1000 python bench.py --no-cpu-baseline

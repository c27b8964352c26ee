#!/bin/bash
run() { # label, env..., groups
  label=$1; shift; g=$1; shift
  env "$@" FSIM_BENCH_FREE_RUN=1 timeout 120 python bench.py --no-cpu-baseline --steps 60 --warmup 2 --groups $g 2>&1 | grep -E "free_run|rror" | cut -c1-120 | sed "s/^/$label g=$g: /"
}
run default 4 A=1
run hwq8 4 GPU_MAX_HW_QUEUES=8
run hwq8 8 GPU_MAX_HW_QUEUES=8
run hwq16 8 GPU_MAX_HW_QUEUES=16
run nodirect 4 AMD_DIRECT_DISPATCH=0
run hwq2 2 GPU_MAX_HW_QUEUES=2
run hwq1 2 GPU_MAX_HW_QUEUES=1

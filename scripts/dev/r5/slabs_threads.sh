export PYTHONPATH=$PWD
O=gpurun_out/r5b; mkdir -p $O
run() { # tag env... -- args
  tag=$1; shift
  env "$@" timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline $ARGS > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("BENCH", sys.argv[2], round(d['value']), "env-steps/s", round(d['ms_per_step'],3), "ms/step kernel", round(d['roofline']['kernel_avg_ms'],3))
except Exception as e: print("BENCH", sys.argv[2], "failed", e)
PY
  grep "^slab" $O/$tag.err
}
ARGS="--groups 4" run g4_thr FSIM_BENCH_THREADS=1 FSIM_BENCH_TRACE=1
ARGS="--groups 8" run g8_q8_thr GPU_MAX_HW_QUEUES=8 FSIM_BENCH_THREADS=1 FSIM_BENCH_TRACE=1
ARGS="--groups 16" run g16_q16_thr GPU_MAX_HW_QUEUES=16 FSIM_BENCH_THREADS=1 FSIM_BENCH_TRACE=1
ARGS="--groups 16 --no-lookahead" run g16_q16_thr_nola GPU_MAX_HW_QUEUES=16 FSIM_BENCH_THREADS=1 FSIM_BENCH_TRACE=1
ARGS="--groups 16 --multi-wave off" run g16_q16_thr_mwoff GPU_MAX_HW_QUEUES=16 FSIM_BENCH_THREADS=1 FSIM_BENCH_TRACE=1
ARGS="--groups 8 --multi-wave off" run g8_q8_thr_mwoff GPU_MAX_HW_QUEUES=8 FSIM_BENCH_THREADS=1 FSIM_BENCH_TRACE=1
ARGS="--groups 4 --multi-wave off" run g4_thr_mwoff FSIM_BENCH_THREADS=1 FSIM_BENCH_TRACE=1

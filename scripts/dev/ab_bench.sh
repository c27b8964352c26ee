#!/bin/bash
# bench-only A/B of library variants on one GPU box: ab_bench.sh <outdir> <tag> ...   (tag -> furniture_amd/csrc/libfsim_<tag>.so; "base" = libfsim.so)
O=gpurun_out/$1; mkdir -p $O; shift
for rep in 1 2; do
for tag in "$@"; do
  lib=furniture_amd/csrc/libfsim_$tag.so; [ "$tag" = base ] && lib=furniture_amd/csrc/libfsim.so
  FSIM_LIB=$PWD/$lib timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/bench_${tag}_$rep.json 2> $O/bench_${tag}_$rep.err
  python - $O/bench_${tag}_$rep.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("BENCH", sys.argv[2], round(d['value']), "env-steps/s", round(d['ms_per_step'],3), "ms/step kernel", round(d['roofline']['kernel_avg_ms'],3))
except Exception as e: print("BENCH", sys.argv[2], "failed", e)
PY
done
done

#!/bin/bash
# A/B two library variants on the GPU box: ab_libs.sh <outdir> <tagA> <tagB> ...   (tag -> furniture_amd/csrc/libfsim_<tag>.so and ..._<tag>_prof.so; "base" = libfsim.so)
O=gpurun_out/$1; mkdir -p $O; shift
for tag in "$@"; do
  lib=furniture_amd/csrc/libfsim_$tag.so; prof=furniture_amd/csrc/libfsim_${tag}_prof.so
  [ "$tag" = base ] && lib=furniture_amd/csrc/libfsim.so && prof=furniture_amd/csrc/libfsim_prof.so
  FSIM_LIB=$PWD/$prof timeout 200 python scripts/gpu_phase_profile.py 7 > $O/phase_$tag.log 2>&1
  echo "== $tag"; grep -E "^step  [05]|per-substep kcycles \(median|SLOW env [0-9]+: Mcyc" $O/phase_$tag.log | cut -c1-460
  for rep in 1 2; do
    FSIM_LIB=$PWD/$lib timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/bench_${tag}_$rep.json 2> $O/bench_${tag}_$rep.err
    python - $O/bench_${tag}_$rep.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("BENCH", sys.argv[2], round(d['value']), "env-steps/s", round(d['ms_per_step'],3), "ms/step kernel", round(d['roofline']['kernel_avg_ms'],3))
except Exception as e: print("BENCH", sys.argv[2], "failed", e)
PY
  done
done

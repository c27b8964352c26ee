#!/bin/bash
# round-2 A/B helper: $1 = output dir under gpurun_out, remaining args = variant tags (env files under scripts/dev/variants/*.env optional)
O=gpurun_out/$1; mkdir -p $O
run() { # tag, env assignments...
  tag=$1; shift
  env "$@" timeout 200 python scripts/gpu_phase_profile.py 10 > $O/phase_$tag.log 2>&1
  grep -E "^step  [059]|per-substep|SLOW|Hessian split" $O/phase_$tag.log | cut -c1-700
  env "$@" timeout 200 python bench.py --steps 60 --warmup 2 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - $O/bench_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("BENCH", sys.argv[2], round(d['value']), "env-steps/s", round(d['ms_per_step'],3), "ms/step kernel", round(d['roofline']['kernel_avg_ms'],3), d['config'].get('kernel_variant'))
except Exception as e: print("BENCH", sys.argv[2], "failed", e)
PY
}

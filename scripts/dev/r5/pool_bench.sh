# development (round 5): bench.py with the shared work pool against the launch-per-slab-step path, over slab counts.  usage: pool_bench.sh <outdir> [steps] [warmup]
export PYTHONPATH=$PWD
O=gpurun_out/$1; mkdir -p $O; ST=${2:-60}; WU=${3:-5}
run() { # tag "args" env...
  tag=$1; a=$2; shift; shift
  env "$@" timeout 60 python bench.py --steps $ST --warmup $WU --no-cpu-baseline $a > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("BENCH", sys.argv[2], round(d['value']), "env-steps/s", round(d['ms_per_step'],3), "ms/step kernel", round(d['roofline']['kernel_avg_ms'],3), "lat", d['roofline'].get('slab_step_latency_ms'), d['config'].get('work_pool'), "dropped", d['config']['envs_that_dropped_contacts'], "resteps", d['config']['overflow_resteps'], "finite", d['config']['obs_finite'])
except Exception as e: print("BENCH", sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}

#!/bin/bash
# round 4, GPU call A: the whole GPU suite (new: look-ahead reset, fsim_step determinism, kernel-path parametrisation) + bench A/B
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4a; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=15 > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
tail -40 $O/gpu_tests.txt | grep -E "passed|failed|FAILED|ERROR|rc=" | head -30
B="python bench.py --no-cpu-baseline"
timeout 300 $B --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
timeout 300 $B --steps 20 --warmup 5 --no-lookahead > $O/bench_20_5_nola.json 2> $O/bench_20_5_nola.err
FSIM_BENCH_TRACE=1 timeout 400 $B > $O/bench_default.json 2> $O/bench_default.err
timeout 400 $B --no-lookahead > $O/bench_default_nola.json 2> $O/bench_default_nola.err
FSIM_LA_DEFER=0 timeout 300 $B --steps 20 --warmup 5 > $O/bench_20_5_defer0.json 2> $O/bench_20_5_defer0.err
FSIM_LA_DEFER=0 timeout 400 $B > $O/bench_default_defer0.json 2> $O/bench_default_defer0.err
timeout 300 $B --steps 100 --warmup 10 > $O/bench_100_10.json 2> $O/bench_100_10.err
for f in $O/bench_*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); c=d['config']; print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', d['roofline']['kernel'], 'kms', round(d['roofline']['kernel_avg_ms'],3), 'swapped', c['resets_taken_from_lookahead'], 'inline', c['resets_inside_step_launch'], 'launched', c['lookahead_resets_launched_in_timed_region'])" 2>&1 | tail -1)"; done

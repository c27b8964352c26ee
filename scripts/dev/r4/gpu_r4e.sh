#!/bin/bash
# round 4, GPU call E: whole GPU suite per file + perf A/B (matrix-core assembly vs LDS assembly) + phase profile
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4e; mkdir -p $O
cd $R
for f in $(grep -l mark.gpu tests/test_*.py); do
  b=$(basename $f .py)
  timeout 600 python -m pytest $f -m gpu -q -rfE --tb=short > $O/$b.txt 2>&1; rc=$?
  echo "$b rc=$rc: $(grep -E "passed|failed|error" $O/$b.txt | tail -1)"
  [ $rc -ne 0 ] && [ $rc -ne 5 ] && grep -E "^(FAILED|ERROR|E  )" $O/$b.txt | head -12
done
B="python bench.py --no-cpu-baseline"
for lib in libfsim.so libfsim_nomfma.so; do
  for v in "20_5:--steps 20 --warmup 5" "100_10:--steps 100 --warmup 10"; do
    n=${v%%:*}; a=${v#*:}
    FSIM_LIB=$R/furniture_amd/csrc/$lib timeout 400 $B $a --no-lookahead > $O/bench_${lib%.so}_${n}_nola.json 2> $O/bench_${lib%.so}_${n}_nola.err
  done
  FSIM_LIB=$R/furniture_amd/csrc/$lib timeout 400 $B > $O/bench_${lib%.so}_default.json 2> $O/bench_${lib%.so}_default.err
done
for f in $O/*bench_*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); c=d['config']; print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', 'kms', round(d['roofline'].get('kernel_avg_ms'),3), 'swapped', c.get('resets_taken_from_lookahead'), 'inline', c.get('resets_inside_step_launch'), 'units', c.get('lookahead_reset_units_in_timed_region'))" 2>&1 | tail -1)"; done
FSIM_MW=0 FSIM_PROF_N=1024 timeout 300 python $R/scripts/gpu_phase_profile.py 8 > $O/phase_onewave_1024.txt 2>&1
grep -n "SLOW env" $O/phase_onewave_1024.txt | cut -c1-400

#!/bin/bash
# round 4, GPU call C: abort triage (multi-wave workgroup re-use), grevback diagnostics, full GPU suite per file, bench A/B incl. the round-3 library
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c; mkdir -p $O
cd $R
export AMD_LOG_LEVEL=1
for c in "2 ik" "6 ik" "6 impedance"; do
  for k in "" "FSIM_MW_K=0" "FSIM_MW=0" "FSIM_MW=all"; do
    echo "== case $c  $k" >> $O/triage.txt
    env $k timeout 120 python scripts/dev/r4_diag.py ik $c >> $O/triage.txt 2>&1; echo "   rc=$?" >> $O/triage.txt
  done
done
unset AMD_LOG_LEVEL
grep -E "^==|rc=|ok|error|Error|abort" $O/triage.txt | head -60
timeout 300 python scripts/dev/r4_diag.py grev > $O/grev.txt 2>&1; tail -20 $O/grev.txt
for f in $(grep -l mark.gpu tests/test_*.py); do
  b=$(basename $f .py)
  timeout 600 python -m pytest $f -m gpu -q -rfE --tb=short > $O/$b.txt 2>&1; rc=$?
  echo "$b rc=$rc: $(grep -E "passed|failed|error" $O/$b.txt | tail -1)"
  [ $rc -ne 0 ] && [ $rc -ne 5 ] && grep -E "^(FAILED|ERROR|E  )" $O/$b.txt | head -16
done
B="python bench.py --no-cpu-baseline"
for v in "20_5:--steps 20 --warmup 5" "100_10:--steps 100 --warmup 10" "default:"; do
  n=${v%%:*}; a=${v#*:}
  timeout 400 $B $a > $O/bench_$n.json 2> $O/bench_$n.err
  timeout 400 $B $a --no-lookahead > $O/bench_${n}_nola.json 2> $O/bench_${n}_nola.err
done
FSIM_LIB=$R/furniture_amd/csrc/libfsim_nomfma.so timeout 300 $B --steps 20 --warmup 5 > $O/bench_20_5_nomfma.json 2> $O/bench_20_5_nomfma.err
FSIM_LIB=$R/furniture_amd/csrc/libfsim_nomfma.so timeout 300 $B --steps 100 --warmup 10 > $O/bench_100_10_nomfma.json 2> $O/bench_100_10_nomfma.err
FSIM_MW=0 FSIM_PROF_N=1024 timeout 300 python scripts/gpu_phase_profile.py 8 > $O/phase_onewave_1024.txt 2>&1
FSIM_PROF_N=1024 timeout 300 python scripts/gpu_phase_profile.py 8 > $O/phase_rule_1024.txt 2>&1
(cd .r3ab && timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/r3_bench_20_5.json 2> $O/r3_bench_20_5.err; timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 > $O/r3_bench_100_10.json 2> $O/r3_bench_100_10.err)
timeout 300 $B --steps 20 --warmup 5 > $O/bench_20_5_again.json 2> $O/bench_20_5_again.err
for f in $O/*bench_*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); c=d['config']; print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', 'kms', d['roofline'].get('kernel_avg_ms'), 'swapped', c.get('resets_taken_from_lookahead'), 'inline', c.get('resets_inside_step_launch'), 'units', c.get('lookahead_reset_units_in_timed_region'))" 2>&1 | tail -1)"; done

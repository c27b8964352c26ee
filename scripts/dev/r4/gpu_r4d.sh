#!/bin/bash
# round 4, GPU call D: rocgdb on the generic multi-wave abort, perf A/B of library variants with kernel traces, targeted tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4d; mkdir -p $O
cd $R
echo "== generic impedance, every env multi-wave" > $O/triage.txt
FSIM_GENERIC=1 FSIM_MW_K=0 AMD_LOG_LEVEL=1 timeout 120 python scripts/dev/r4_diag.py ik 6 impedance >> $O/triage.txt 2>&1; echo "   rc=$?" >> $O/triage.txt
echo "== rocgdb: ik, every env multi-wave" >> $O/triage.txt
FSIM_MW_K=0 timeout 300 rocgdb --batch -ex "set pagination off" -ex "set confirm off" -ex run -ex "info threads" -ex bt -ex "x/24i \$pc-48" -ex "info registers pc exec" --args python scripts/dev/r4_diag.py ik 2 ik > $O/rocgdb_ik.txt 2>&1
grep -n "received signal\|in k_env\|in fs_\|in env_\|in mw_\|=> \|Thread.*stopped\|fault" $O/rocgdb_ik.txt | head -20 >> $O/triage.txt
cat $O/triage.txt | tail -40
for f in tests/test_lookahead_gpu.py tests/test_determinism_gpu.py tests/test_dense_gpu.py tests/test_all_furniture_gpu.py tests/test_gpu_parity.py tests/test_contact_stress_gpu.py tests/test_vec_env_gpu.py; do
  b=$(basename $f .py)
  timeout 600 python -m pytest $f -m gpu -q -rfE --tb=short > $O/$b.txt 2>&1; rc=$?
  echo "$b rc=$rc: $(grep -E "passed|failed|error" $O/$b.txt | tail -1)"
  [ $rc -ne 0 ] && [ $rc -ne 5 ] && grep -E "^(FAILED|ERROR|E  )" $O/$b.txt | head -12
done
B="python bench.py --no-cpu-baseline"
for lib in libfsim.so libfsim_nomfma.so libfsim_big17.so; do
  for v in "20_5:--steps 20 --warmup 5" "100_10:--steps 100 --warmup 10"; do
    n=${v%%:*}; a=${v#*:}
    FSIM_LIB=$R/furniture_amd/csrc/$lib timeout 400 $B $a --no-lookahead > $O/bench_${lib%.so}_${n}_nola.json 2> $O/bench_${lib%.so}_${n}_nola.err
  done
done
timeout 400 $B > $O/bench_default.json 2> $O/bench_default.err
timeout 400 $B --steps 100 --warmup 10 > $O/bench_100_10.json 2> $O/bench_100_10.err
(cd .r3ab && timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/r3_bench_20_5.json 2> $O/r3_bench_20_5.err; timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 > $O/r3_bench_100_10.json 2> $O/r3_bench_100_10.err)
for f in $O/*bench_*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); c=d['config']; print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', 'kms', d['roofline'].get('kernel_avg_ms'), 'swapped', c.get('resets_taken_from_lookahead'), 'inline', c.get('resets_inside_step_launch'), 'units', c.get('lookahead_reset_units_in_timed_region'))" 2>&1 | tail -1)"; done
# kernel traces: round-3 library vs this one (same workload, 30 steps)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_r4 -o kt -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-lookahead > $O/kt_r4.log 2>&1
(cd $R/.r3ab && timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_r3 -o kt -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/kt_r3.log 2>&1)
for t in r4 r3; do DB=$(find $O/kt_$t -name "*.db" | head -1); [ -n "$DB" ] && python $R/scripts/rocprof_summary.py $DB $O/kt_${t}_summary.txt "bench.py --steps 30 --warmup 5 ($t library)" | tail -8; done
rm -rf $O/kt_r4 $O/kt_r3
FSIM_MW=0 FSIM_PROF_N=1024 timeout 300 python $R/scripts/gpu_phase_profile.py 8 > $O/phase_onewave_1024.txt 2>&1

#!/bin/bash
# round 4, GPU call F: perf A/B of four library variants + grevback substep diagnostic + phase profile
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4f; mkdir -p $O
cd $R
timeout 200 python scripts/dev/r4_diag.py grevsub > $O/grevsub.txt 2>&1; head -30 $O/grevsub.txt | cut -c1-260
B="python bench.py --no-cpu-baseline"
for lib in libfsim.so libfsim_nomfma_op.so libfsim_nomfma.so libfsim_mfma_noop.so; do
  for v in "20_5:--steps 20 --warmup 5" "100_10:--steps 100 --warmup 10"; do
    n=${v%%:*}; a=${v#*:}
    FSIM_LIB=$R/furniture_amd/csrc/$lib timeout 400 $B $a --no-lookahead > $O/bench_${lib%.so}_${n}_nola.json 2> $O/bench_${lib%.so}_${n}_nola.err
  done
done
for lib in libfsim.so libfsim_nomfma_op.so; do
  FSIM_LIB=$R/furniture_amd/csrc/$lib timeout 400 $B > $O/bench_${lib%.so}_default.json 2> $O/bench_${lib%.so}_default.err
done
for f in $O/*bench_*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); c=d['config']; print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', 'kms', round(d['roofline'].get('kernel_avg_ms'),3), 'swapped', c.get('resets_taken_from_lookahead'), 'inline', c.get('resets_inside_step_launch'))" 2>&1 | tail -1)"; done
FSIM_MW=0 FSIM_PROF_N=1024 timeout 300 python $R/scripts/gpu_phase_profile.py 8 > $O/phase_onewave_1024.txt 2>&1
grep -n "SLOW env\|^step  5" $O/phase_onewave_1024.txt | cut -c1-400
timeout 300 python -m pytest tests/test_dense_gpu.py tests/test_gpu_parity.py tests/test_contact_stress_gpu.py -m gpu -q 2>&1 | tail -5

#!/bin/bash
# round 6, third GPU call: GPU suite on the committed library; A/B of the team projection without the composite pass (libfsim_A*.so)
O=gpurun_out/r6c; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
# correctness of variant A: parity / determinism / stress / episodes files against the same library path
( FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_A.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_determinism_gpu.py tests/test_contact_stress_gpu.py tests/test_capi_cpu.py tests/test_lds_clean_gpu.py -m gpu -x -q ) > $O/pytest_A.txt 2>&1
tail -3 $O/pytest_A.txt
( FSIM_MW_K=0 FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_A.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_capi_cpu.py -m gpu -x -q ) > $O/pytest_A_k0.txt 2>&1
tail -3 $O/pytest_A_k0.txt
for tag in r6base A; do
  FSIM_MW=all FSIM_PROF_N=1024 FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_${tag}_prof.so timeout 300 python scripts/gpu_phase_profile.py 7 > $O/phase_mwall_$tag.txt 2>&1
  FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_${tag}_prof.so timeout 300 python scripts/gpu_phase_profile.py 7 > $O/phase_rule_$tag.txt 2>&1
  echo "== $tag (every env on a team)"; grep -E "SLOW env [0-9]+: Mcyc|multi-wave iteration" $O/phase_mwall_$tag.txt | cut -c1-420 | head -8
done
for rep in 1 2 3; do
  for tag in r6base A; do
    FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_$tag.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/bench_${tag}_$rep.json 2> $O/bench_${tag}_$rep.err
    echo BENCH $tag $rep $(python -c "import json; d=json.load(open('$O/bench_${tag}_$rep.json')); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_avg_ms'],3))")
  done
done
for tag in r6base A; do
  FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_$tag.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --episode-window 0 > $O/bench20_${tag}.json 2> $O/bench20_${tag}.err
  echo BENCH20 $tag $(python -c "import json; d=json.load(open('$O/bench20_${tag}.json')); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_avg_ms'],3))")
done

#!/bin/bash
# end of round 4, final library: whole GPU suite per file, the opt-in matrix-core build through the parity tests, profiles
R=$PWD
export PYTHONPATH=$R
rm -rf gpurun_out/gpu_tests_final; bash scripts/gpu_tests_per_file.sh gpurun_out/gpu_tests_final > gpurun_out/gpu_tests_final.log 2>&1
mkdir -p gpurun_out/r4t
FSIM_LIB=$R/furniture_amd/csrc/libfsim_mfma.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_contact_stress_gpu.py tests/test_determinism_gpu.py -q -m gpu 2>&1 | tail -4 > gpurun_out/r4t/mfma_optin_tests.txt
sed -i 's#gpurun_out/r4q#gpurun_out/r4u#g; s#r4q_prof#r4u_prof#g' scripts/dev/r4/gpu_r4q.sh
bash scripts/dev/r4/gpu_r4q.sh > gpurun_out/r4t/r4q.log 2>&1
cat gpurun_out/gpu_tests_final.log; cat gpurun_out/r4t/mfma_optin_tests.txt; tail -12 gpurun_out/r4t/r4q.log

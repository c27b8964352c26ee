#!/bin/bash
# round 6, the library as the round ends (+ huge islands routed to the ladder): the profile set again + the GPU suite in its variant modes + bench lines
bash scripts/profile_round.sh r6z > gpurun_out/r6z_profile.log 2>&1
tail -12 gpurun_out/r6z_profile.log
O=gpurun_out/r6z; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 600 $O/bench_driver.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
( time FSIM_MW_K=0 timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_mwk0.txt 2>&1; tail -3 $O/pytest_mwk0.txt
( time FSIM_TEST_POISON=7fc00000 timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_poison.txt 2>&1; tail -3 $O/pytest_poison.txt
( time FSIM_GENERIC=1 timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_generic.txt 2>&1; tail -3 $O/pytest_generic.txt

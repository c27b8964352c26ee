#!/bin/bash
# round 6, first GPU call: divergence control (1024 x 152), baseline bench lines of the round-5 library on this box, GPU suite
mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
( time python scripts/divergence_control.py 1024 150 152 ) > gpurun_out/r6a/divergence_control.txt 2> gpurun_out/r6a/divergence_control.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6a/bench_20_5.json 2> gpurun_out/r6a/bench_20_5.err
python bench.py --gpus 1 --steps 100 --warmup 10 > gpurun_out/r6a/bench_100_10.json 2> gpurun_out/r6a/bench_100_10.err
python bench.py --gpus 1 --steps 100 --warmup 10 --threads 1 > gpurun_out/r6a/bench_100_10_threads.json 2> gpurun_out/r6a/bench_100_10_threads.err
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r6a/pytest_gpu.txt 2>&1
tail -3 gpurun_out/r6a/pytest_gpu.txt
tail -4 gpurun_out/r6a/divergence_control.txt
cat gpurun_out/r6a/bench_*.json | cut -c1-400

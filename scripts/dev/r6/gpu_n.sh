#!/bin/bash
# round 6: the 512-slot rung -- its tests, the sweeps whose exception lists it changes, and an A/B of the benchmark (the island map's words were repacked)
O=gpurun_out/r6n; mkdir -p $O
make -C oracle -s
timeout 1500 python -m pytest tests/test_overflow_restep_gpu.py tests/test_lookahead_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/tests_a.txt
timeout 2400 python -m pytest tests/test_all_furniture_gpu.py tests/test_agents_catalogue_gpu.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -30 > $O/tests_b.txt
for i in 1 2 3; do
  FSIM_LIB=$PWD/scripts/dev/r6/libfsim_head.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head', d['value'], d['ms_per_step'])" >> $O/ab.txt
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
cat $O/tests_a.txt $O/tests_b.txt $O/ab.txt

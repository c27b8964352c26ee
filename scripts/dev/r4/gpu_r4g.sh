#!/bin/bash
# round 4 call G: who finishes last in the rule's kernel; grevback substep diagnostic; dense test with the default library
O=gpurun_out/r4g; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python scripts/dev/r4_diag.py grevsub > $O/grevsub.txt 2>&1
FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_tl.so timeout 300 python scripts/dev/timeline_x.py 30 auto > $O/timeline_rule.txt 2>&1
FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_tl.so timeout 300 python scripts/dev/timeline_x.py 30 off > $O/timeline_off.txt 2>&1
timeout 300 python -m pytest tests/test_dense_gpu.py -x -q 2>&1 | tail -15 > $O/test_dense.txt
python bench.py --no-lookahead --steps 20 --warmup 5 > $O/bench_20_5_nola.json 2> $O/bench_20_5_nola.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -5 $O/grevsub.txt; tail -4 $O/timeline_rule.txt; tail -4 $O/test_dense.txt

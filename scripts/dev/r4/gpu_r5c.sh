#!/bin/bash
# final library (opaque thread index in the persistent loops, n/4 + n/16 workgroups): GPU suite in one process, smoke, profiles
O=gpurun_out/r5c; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite_one_process.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_suite_one_process.txt
tail -3 $O/gpu_suite_one_process.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
bash scripts/dev/r4/gpu_r4q.sh > $O/r4q.log 2>&1; tail -8 $O/r4q.log

#!/bin/bash
O=gpurun_out/r5i; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite_one_process.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_suite_one_process.txt
tail -3 $O/gpu_suite_one_process.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt

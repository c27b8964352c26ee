#!/bin/bash
O=gpurun_out/r4p; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_b1_residuals_gpu.py -x -q 2>&1 | tail -30 > $O/b1.txt; tail -30 $O/b1.txt

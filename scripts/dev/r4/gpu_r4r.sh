#!/bin/bash
O=gpurun_out/r4r; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_b1_residuals_gpu.py -q -m gpu 2>&1 | tail -30 > $O/tests.txt; tail -30 $O/tests.txt

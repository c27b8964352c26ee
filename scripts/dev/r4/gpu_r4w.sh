#!/bin/bash
O=gpurun_out/r4w; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_b1_residuals_gpu.py -q -m gpu -x 2>&1 | tail -25 > $O/b1.txt; tail -25 $O/b1.txt

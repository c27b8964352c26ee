#!/bin/bash
O=gpurun_out/r5g; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_overflow_restep_gpu.py -q -m gpu -x -s 2>&1 | tail -25 > $O/t.txt; tail -25 $O/t.txt

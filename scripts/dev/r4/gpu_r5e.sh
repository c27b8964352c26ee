#!/bin/bash
O=gpurun_out/r5e; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python scripts/dev/overflow_census.py 4096 450 > $O/census.txt 2>&1; tail -3 $O/census.txt

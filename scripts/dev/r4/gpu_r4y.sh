#!/bin/bash
O=gpurun_out/r4y; mkdir -p $O
export PYTHONPATH=$PWD
FSIM_PROF_N=1024 FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_tl.so timeout 300 python scripts/dev/timeline_x.py 12 rule > $O/timeline_1024.txt 2>&1
grep "^step" $O/timeline_1024.txt | cut -c1-120,300-520 | tail -6

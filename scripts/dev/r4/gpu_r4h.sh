#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
export PYTHONPATH=$PWD
FSIM_TL_DUMP=$O/tl_rule.npy FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_tl.so timeout 300 python scripts/dev/timeline_x.py 50 rule > $O/timeline_rule.txt 2>&1
tail -3 $O/timeline_rule.txt | cut -c1-300

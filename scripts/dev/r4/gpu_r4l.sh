#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python scripts/dev/r4_diag.py grevsub bookcase_grevback_0484 > $O/grevsub.txt 2>&1
timeout 1200 python -m pytest tests/test_all_furniture_gpu.py -x -q 2>&1 | tail -15 > $O/test_all_furniture.txt
sed -n 12,20p $O/grevsub.txt | cut -c1-330; tail -5 $O/test_all_furniture.txt

#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python scripts/dev/release_diag.py 4 5 > $O/release.txt 2>&1
head -5 $O/release.txt | cut -c1-400; grep -c "sub" $O/release.txt

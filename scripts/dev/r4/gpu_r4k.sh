#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python scripts/dev/r4_diag.py grevsub bookcase_grevback_0484 > $O/grevsub.txt 2>&1
FSIM_NCON_MAX=64 timeout 300 python scripts/dev/r4_diag.py grevsub bookcase_grevback_0484 > $O/grevsub_64.txt 2>&1
timeout 300 python scripts/dev/r4_diag.py grevsub bed_dalselv_0270 > $O/bedsub.txt 2>&1
head -22 $O/grevsub.txt | cut -c1-400; echo; sed -n 12,22p $O/grevsub_64.txt | cut -c1-300; echo; sed -n 1,3p $O/bedsub.txt; sed -n 12,20p $O/bedsub.txt | cut -c1-300

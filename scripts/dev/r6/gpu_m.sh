#!/bin/bash
export FSIM_LIB=$PWD/furniture_amd/csrc/libfsim_big.so
make -C oracle -s
timeout 900 python scripts/dev/r6/big_models.py table_liden_0921 bookcase_billy_0191 2>&1 | grep -v amdgpu.ids | tail -40

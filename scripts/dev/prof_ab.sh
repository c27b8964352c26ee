#!/bin/bash
# phase profile of lib variants on identical trajectories (numerically identical variants only)
for v in "$@"; do
  lib=furniture_amd/csrc/libfsim_$v.so
  echo "== $v"
  FSIM_LIB=$PWD/$lib timeout 200 python scripts/gpu_phase_profile.py 7 2>&1 | grep -E "median|SLOW" | tail -4 | sed -E 's/kinematics.*solve:setup 0.0//' | cut -c1-260
done

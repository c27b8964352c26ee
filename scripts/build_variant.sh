#!/bin/bash
# development: builds furniture_amd/csrc/libfsim_<name>.so with extra -D flags.  usage: build_variant.sh name [-DFLAG ...]
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -std=c++17 -shared -fPIC -Wno-unused-value -mllvm -amdgpu-sched-strategy=max-ilp -fno-optimize-sibling-calls "$@" -o furniture_amd/csrc/libfsim_$name.so furniture_amd/csrc/fsim.hip 2>&1 | grep -E "error" -A4 | head -20
ls -la furniture_amd/csrc/libfsim_$name.so

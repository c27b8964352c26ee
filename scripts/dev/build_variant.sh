#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...] -> furniture_amd/csrc/libfsim_NAME.so (development experiments)
cd "$(dirname "$0")/../.." || exit 1
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -std=c++17 -shared -fPIC -Wno-unused-value "$@" -o furniture_amd/csrc/libfsim_$name.so furniture_amd/csrc/fsim.hip 2>&1 | grep -E "error|warning: v" | head
ls -la furniture_amd/csrc/libfsim_$name.so | awk '{print $5, $9}'

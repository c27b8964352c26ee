#!/bin/bash
# builds libfsim.so and the -DFSIM_PROFILE development variant
cd "$(dirname "$0")/.." || exit 1
for f in "" "-DFSIM_PROFILE"; do
  out=libfsim.so; [ -n "$f" ] && out=libfsim_prof.so
  hipcc --offload-arch=gfx950 -O3 -fno-hip-fp32-correctly-rounded-divide-sqrt -std=c++17 -shared -fPIC -Wno-unused-value $f -o furniture_amd/csrc/$out furniture_amd/csrc/fsim.hip 2>&1 | grep -E "error" -A6 | head -20; [ ${PIPESTATUS[0]} -eq 0 ] || echo "BUILD FAILED: $out"
done
ls -la furniture_amd/csrc/*.so | awk '{print $5, $6, $7, $8, $9}'

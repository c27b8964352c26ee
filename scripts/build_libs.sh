#!/bin/bash
# builds libfsim.so and the -DFSIM_PROFILE development variant; fails loudly
cd "$(dirname "$0")/.." || exit 1
rc=0
for f in "" "-DFSIM_PROFILE"; do
  out=libfsim.so; [ -n "$f" ] && out=libfsim_prof.so
  if ! hipcc --offload-arch=gfx950 -O3 -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -std=c++17 -shared -fPIC -Wno-unused-value -mllvm -amdgpu-sched-strategy=max-ilp -fno-optimize-sibling-calls $f -o furniture_amd/csrc/$out.tmp furniture_amd/csrc/fsim.hip > /tmp/build_libs.log 2>&1; then
    echo "BUILD FAILED: $out"; grep -E "error" -A4 /tmp/build_libs.log | head -30; rc=1; rm -f furniture_amd/csrc/$out.tmp
  else
    mv furniture_amd/csrc/$out.tmp furniture_amd/csrc/$out
  fi
done
gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC -o furniture_amd/csrc/libfsim_host.so furniture_amd/csrc/fsim_host.c -lm || rc=1
ls -la furniture_amd/csrc/*.so | awk '{print $5, $6, $7, $8, $9}'
exit $rc

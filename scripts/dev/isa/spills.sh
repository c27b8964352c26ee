#!/bin/bash
# development: where the scratch (spill) instructions of the specialised substep loop sit.  Two clusters of ~112 are the
# prologue / epilogue saves of callee-saved registers (once per call); anything in between is inside the substep loop.
# The register allocation of this 48 k-instruction function is fragile: check after every structural change (no GPU needed).
SRC="$(cd "$(dirname "$0")/../../.." && pwd)/furniture_amd/csrc/fsim.hip"
cd /tmp && hipcc --offload-arch=gfx950 -O3 -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -std=c++17 -Wno-unused-value \
  -mllvm -amdgpu-sched-strategy=max-ilp -fno-optimize-sibling-calls -S --cuda-device-only -o /tmp/fsim_spills.s "$SRC" 2>/dev/null || exit 1
s=$(grep -n "^_Z13fs_substeps_tILb0E7SpecCtx" /tmp/fsim_spills.s | head -1 | cut -d: -f1)
e=$(awk -v s=$s 'NR>s && /^\.Lfunc_end/ {print NR; exit}' /tmp/fsim_spills.s)
sed -n "${s},${e}p" /tmp/fsim_spills.s > /tmp/fsim_sub.s
echo "fs_substeps<false, Spec>: $(grep -c '^\s[a-z]' /tmp/fsim_sub.s) instructions, scratch clusters (first-last line (count)):"
grep -n "scratch_" /tmp/fsim_sub.s | awk -F: '{print $1}' | awk 'NR==1{s=$1;p=$1;n=1;next} {if($1-p>200){print s"-"p" ("n")"; s=$1;n=0} p=$1;n++} END{print s"-"p" ("n")"}' | tr '\n' ' '; echo

#!/bin/bash
# round 6 (third session), after gpu_w.sh found a slab's kernel 13 % faster alone on the chip (4.11 ms) than as one of four (4.66 ms):
# what is shared?  a) the shader clock while one / four slabs run (sysfs, 10 s windows)   b) instruction-cache and scalar-cache counters
R=$PWD; O=$R/gpurun_out/r6w2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ls /sys/class/drm/ > $O/drm.txt 2>&1
SCLK=$(ls /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -1)
echo "sclk file: $SCLK"; cat $SCLK 2>&1 | head -5
watch_clk() { while true; do grep '\*' $SCLK 2>/dev/null | tr '\n' ' '; rocm-smi --showpower 2>/dev/null | grep -i 'power' | head -1 | tr -s ' ' | cut -c1-90; sleep 1.0; done; }
B="python $R/bench.py --steps 2000 --warmup 10 --no-cpu-baseline --episode-window 0"
for mode in "solo --envs-per-gpu 1024 --groups 1" "four"; do
  set -- $mode; tag=$1; shift
  watch_clk > $O/clk_$tag.txt & W=$!
  timeout 300 $B "$@" > $O/long_$tag.json 2> $O/long_$tag.err
  kill $W; wait $W 2>/dev/null
  python -c "
import json
for l in open('$O/long_$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$tag 2000 steps: %.0f env-steps/s  %.3f ms/step  kernel %.3f ms' % (d['value'], d['ms_per_step'], r['kernel_avg_ms']))
"
  echo "clock samples ($tag):"; sort $O/clk_$tag.txt | uniq -c | sort -rn | head -6
done
PB="python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --episode-window 0"
i=0
for set in "GRBM_GUI_ACTIVE SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "GRBM_GUI_ACTIVE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_INPUT_VALID_READYB SQC_TC_REQ SQC_TC_INST_REQ SQC_TC_STALL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -o pmc -- $PB > $O/pmc$i.log 2>&1
  echo "pmc pass $i ($set) rc $?"; grep -i 'error\|invalid\|not found\|unsupported' $O/pmc$i.log | head -3
done
python $R/scripts/pmc_summary.py $O $O/pmc_icache_counters.txt $O/pmc_icache.json > /dev/null 2>&1
cat $O/pmc_icache_counters.txt | cut -c1-200
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4

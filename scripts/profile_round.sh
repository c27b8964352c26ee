#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel-trace stats of bench.py + PMC passes (SQ issue / wait counters, HBM traffic).
#   usage: scripts/profile_round.sh <tag>        -> gpurun_out/<tag>/{kt_*.txt, pmc_*.txt, pmc.json}; copy what is judged to profiles/
# Counters are collected in their own runs (--kernel-trace + --pmc only), FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots).
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-prof}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --episode-window 0"
echo "rocprofv3 --kernel-trace --stats -- $BENCH" > $O/commands.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $BENCH > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
[ -n "$DB" ] && python $R/scripts/rocprof_summary.py $DB $O/kt_two_slabs.txt "bench.py --steps 30 --warmup 5 (default: 4 slabs of 1024 envs on separate streams), Sawyer+table_lack_0825 4096 envs" | tail -6
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt1 -- $BENCH --groups 1 > $O/kt1.log 2>&1
DB=$(find $O/kt1 -name "*.db" | head -1)
[ -n "$DB" ] && python $R/scripts/rocprof_summary.py $DB $O/kt_single_stream.txt "bench.py --steps 30 --warmup 5 --groups 1 (one 4096-env launch per step)" | tail -4
PB="python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --episode-window 0"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "rocprofv3 --kernel-trace --pmc $set -- $PB" >> $O/commands.txt
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -o pmc -- $PB > $O/pmc$i.log 2>&1
done
python $R/scripts/pmc_summary.py $O $O/pmc_sq_counters.txt $O/pmc.json
rm -rf $O/kt $O/kt1 $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4
ls $O

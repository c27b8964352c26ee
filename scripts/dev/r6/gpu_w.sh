#!/bin/bash
# round 6 (third session): does a slab's slowest env run slower because seven other envs share its CU?
#   a) one slab of 1024 envs ALONE on the chip (4 envs per CU, one wave per SIMD) against the same slab as one of four (8 per CU) and one of two,
#      same window (100 steps after 10): ms per step of the solo slab = its kernel + the launch gap, with nobody to wait for
#   b) how busy the CU's one LDS unit is: SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS / SQ_WAIT_INST_LDS against the kernel's cycles
R=$PWD; O=$R/gpurun_out/r6w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0"
line() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$2: %.0f env-steps/s  %.3f ms/step  kernel %.3f ms x %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_launches']))
"; }
for rep in 1 2; do
  timeout 300 $B --envs-per-gpu 1024 --groups 1 > $O/solo_1024_$rep.json 2> $O/solo_1024_$rep.err; line $O/solo_1024_$rep.json solo_1024
  timeout 300 $B --envs-per-gpu 2048 --groups 2 > $O/two_of_1024_$rep.json 2> $O/two_of_1024_$rep.err; line $O/two_of_1024_$rep.json two_x_1024
  timeout 300 $B > $O/four_of_1024_$rep.json 2> $O/four_of_1024_$rep.err; line $O/four_of_1024_$rep.json four_x_1024
done
timeout 300 $B --envs-per-gpu 1024 --groups 1 --multi-wave off > $O/solo_1024_one_wave.json 2> $O/solo_1024_one_wave.err; line $O/solo_1024_one_wave.json solo_1024_one_wave_kernel
timeout 300 $B --envs-per-gpu 1024 --groups 1 --multi-wave all > $O/solo_1024_all_teams.json 2> $O/solo_1024_all_teams.err; line $O/solo_1024_all_teams.json solo_1024_every_env_a_team
PB="python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --episode-window 0"
i=0
for set in "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -o pmc -- $PB > $O/pmc$i.log 2>&1
  echo "pmc pass $i rc $?"; tail -3 $O/pmc$i.log
done
python $R/scripts/pmc_summary.py $O $O/pmc_lds_counters.txt $O/pmc_lds.json > /dev/null 2>&1
cat $O/pmc_lds_counters.txt | cut -c1-200
rm -rf $O/pmc1 $O/pmc2

#!/bin/bash
# round 6 (last experiment): what k_schedule's 0.1 ms are made of.  Kernel traces of the driver-like run with nobody selected for a team (FSIM_MW_K=100000):
#   a) the committed library, look-ahead on      b) look-ahead off (no shadow / serial / episode-length reads)
#   c) look-ahead off + an experiment build (libfsim_ks.so, not kept) whose scheduler does not read E_NITER from the env records (stride ~2 KB: a cache line per env)
R=$PWD; O=$R/gpurun_out/r6ks; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export FSIM_MW_K=100000
run() { tag=$1; shift; env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o kt -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --episode-window 0 > $O/kt_$tag.log 2>&1
  DB=$(find $O/kt_$tag -name "*.db" | head -1); [ -n "$DB" ] && python $R/scripts/rocprof_summary.py $DB $O/kt_$tag.txt "$tag" > /dev/null; grep "k_schedule\|k_env_step_x" $O/kt_$tag.txt | cut -c1-20,60-110 | sed "s/^/$tag: /"; rm -rf $O/kt_$tag; }
run a_lookahead FSIM_LIB=$R/furniture_amd/csrc/libfsim.so
run b_no_lookahead FSIM_LIB=$R/furniture_amd/csrc/libfsim.so FSIM_NO_LOOKAHEAD=1
run c_no_lookahead_no_record_read FSIM_LIB=$R/furniture_amd/csrc/libfsim_ks.so FSIM_NO_LOOKAHEAD=1

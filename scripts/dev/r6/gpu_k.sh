#!/bin/bash
# round 6: k_schedule warming its own code (libfsim_warm.so) against the committed library: kernel trace of both, bench x 3
O=gpurun_out/r6k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in base warm; do
  lib=$R/furniture_amd/csrc/libfsim.so; [ $tag = warm ] && lib=$R/furniture_amd/csrc/libfsim_warm.so
  FSIM_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt_$tag -o kt -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --episode-window 0 > $R/$O/kt_$tag.log 2>&1
  DB=$(find $R/$O/kt_$tag -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/scripts/rocprof_summary.py $DB $R/$O/kt_$tag.txt "bench.py --steps 30 --warmup 5, $tag" | head -8 | tail -4
  rm -rf $R/$O/kt_$tag
done
cd $R
for rep in 1 2 3; do for tag in base warm; do
  lib=$R/furniture_amd/csrc/libfsim.so; [ $tag = warm ] && lib=$R/furniture_amd/csrc/libfsim_warm.so
  FSIM_LIB=$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > $O/bench_${tag}_$rep.json 2> $O/bench_${tag}_$rep.err
  echo BENCH $tag $rep $(python -c "import json; d=json.load(open('$O/bench_${tag}_$rep.json')); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_avg_ms'],3))")
done; done

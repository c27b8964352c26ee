#!/bin/bash
# HBM traffic of the step kernel only: the two TCC passes of scripts/profile_round.sh (FETCH_SIZE, WRITE_SIZE in separate runs)
#   usage: scripts/pmc_traffic.sh <tag>   -> gpurun_out/<tag>/pmc.json
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-traffic}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PB="python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --episode-window 0"
i=2
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -o pmc -- $PB > $O/pmc$i.log 2>&1
done
python $R/scripts/pmc_summary.py $O $O/pmc_sq_counters.txt $O/pmc.json 2>&1 | tail -3
rm -rf $O/pmc3 $O/pmc4

#!/bin/bash
# development: same-box A/B of the wave-priority experiment (-DFSIM_PRIO / -DFSIM_PRIO_LOOP builds beside the default library)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prio; mkdir -p $O
for rep in 1 2 3; do
  for v in "" _prio1 _prio2; do
    FSIM_LIB=$R/furniture_amd/csrc/libfsim$v.so timeout 120 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 2>/dev/null | tail -1 > $O/b$v.$rep.json
    python -c "
import json; d=json.load(open('$O/b$v.$rep.json')); print('lib$v rep $rep', round(d['value']), d['ms_per_step'])"
  done
done

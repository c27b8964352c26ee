#!/bin/bash
# soak: 3100 steps of the benchmark workload (20 batch-wide resets, look-ahead, overflow re-steps), and the other kernel modes for 600 steps
O=gpurun_out/r5l; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python bench.py --steps 3000 --warmup 100 --no-cpu-baseline > $O/soak_3000.json 2> $O/soak.err
timeout 300 python bench.py --multi-wave off --steps 500 --warmup 100 --no-cpu-baseline > $O/off_500.json 2> $O/off.err
timeout 300 python bench.py --multi-wave all --steps 200 --warmup 50 --no-cpu-baseline --no-lookahead > $O/all_200.json 2> $O/all.err
python - <<'PY'
import json
for f in ("soak_3000","off_500","all_200"):
    try:
        d=json.loads(open("gpurun_out/r5l/%s.json"%f).read().strip().splitlines()[-1]); c=d["config"]
        print(f, round(d["value"]), "finite", c["obs_finite"], "resets", c["resets_in_timed_region"], "from lookahead", c["resets_taken_from_lookahead"], "resteps", c["overflow_resteps"], "dropped", c["envs_that_dropped_contacts"], d["roofline"]["kernel"])
    except Exception as e: print(f,"ERR",e, open("gpurun_out/r5l/%s.err"%f.split("_")[0]).read()[-300:])
PY

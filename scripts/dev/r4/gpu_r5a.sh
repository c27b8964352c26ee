#!/bin/bash
# the step kernel leaves the next launch's schedule behind: tests, then A/B against the separate k_schedule launch (same library)
O=gpurun_out/r5a; mkdir -p $O
export PYTHONPATH=$PWD
for t in test_determinism_gpu test_lookahead_gpu test_gpu_parity test_contact_stress_gpu test_vec_env_gpu test_baseline_configs_gpu; do
  timeout 900 python -m pytest tests/$t.py -q -m gpu -x 2>&1 | tail -3 > $O/$t.txt; echo "$t: $(tail -1 $O/$t.txt)"
done
for k in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/in${k}_20_5.json 2> $O/err.txt
  FSIM_NO_INKERNEL_SCHED=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/sep${k}_20_5.json 2> $O/err.txt
  python bench.py --no-lookahead --steps 100 --warmup 10 --no-cpu-baseline > $O/in${k}_100_10.json 2> $O/err.txt
  FSIM_NO_INKERNEL_SCHED=1 python bench.py --no-lookahead --steps 100 --warmup 10 --no-cpu-baseline > $O/sep${k}_100_10.json 2> $O/err.txt
done
python bench.py --no-cpu-baseline > $O/in_default.json 2> $O/err.txt
FSIM_NO_INKERNEL_SCHED=1 python bench.py --no-cpu-baseline > $O/sep_default.json 2> $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5a/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), round(d["roofline"]["kernel_avg_ms"],3))
    except Exception as e: print(f, "ERR", e)
PY

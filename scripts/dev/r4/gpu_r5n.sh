#!/bin/bash
# last call of the round: what the driver runs (suite in one process, smoke, the driver's bench command) plus the 1-rank RCCL launch
O=gpurun_out/r5n; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite_one_process.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_suite_one_process.txt
tail -2 $O/gpu_suite_one_process.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench.err
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_torchrun.json 2> $O/bench_torchrun.err
python - <<'PY'
import json
for f in ("bench_driver_command","bench_torchrun"):
    try: d=json.loads(open("gpurun_out/r5n/%s.json"%f).read().strip().splitlines()[-1]); print(f, round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"] if "cpu_baseline" in d else None, d["config"]["rccl_world"])
    except Exception as e: print(f,"ERR",e)
PY

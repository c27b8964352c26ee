#!/bin/bash
# round 4, end of round: rocprofv3 kernel trace + PMC passes, the three bench protocols, the 1-rank RCCL launch, timeline, phase profile
R=$PWD; O=$R/gpurun_out/r4q; mkdir -p $O
export PYTHONPATH=$R
python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench_steps20_warmup5.err
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_steps100_warmup10.json 2> $O/bench_steps100_warmup10.err
python bench.py --no-cpu-baseline > $O/bench_default_1000_100.json 2> $O/bench_default.err
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_torchrun_1rank_rccl.json 2> $O/bench_torchrun.err
bash scripts/profile_round.sh r4q_prof > $O/profile_round.log 2>&1
FSIM_LIB=$R/furniture_amd/csrc/libfsim_tl.so timeout 300 python scripts/dev/timeline_x.py 30 rule > $O/timeline_rule_4096.txt 2>&1
FSIM_LIB=$R/furniture_amd/csrc/libfsim_prof.so FSIM_MW=0 FSIM_PROF_N=1024 timeout 300 python scripts/gpu_phase_profile.py 8 > $O/phase_profile_one_wave_1024.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4q/bench_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d.get("rccl_world"))
    except Exception as e: print(f, "ERR", e)
PY
tail -5 $O/profile_round.log; ls gpurun_out/r4q_prof

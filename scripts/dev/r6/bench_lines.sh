# development (round 6): the bench lines kept under profiles/ -- the contract's command, the SURVEY 8(d) protocol, the RCCL branch with one
# rank, round robin against one host thread per slab, the other BASELINE configs / widened rows.  usage: bench_lines.sh <outdir>
export PYTHONPATH=$PWD
O=gpurun_out/$1; mkdir -p $O
b() { tag=$1; shift; timeout 600 "$@" > $O/$tag.json 2> $O/$tag.err; python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['value']), "env-steps/s", round(d['ms_per_step'],3), "ms/step", (d['config'].get('episode_window') or {}).get('env_steps_per_s'))
except Exception as e: print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json','.err')).read()[-500:])
PY
}
b bench_driver_command_steps20_warmup5 python bench.py --gpus 1 --steps 20 --warmup 5
b bench_driver_command_steps20_warmup5_repeat python bench.py --gpus 1 --steps 20 --warmup 5
b bench_default_1000_100 python bench.py --no-cpu-baseline
b bench_steps100_warmup10 python bench.py --steps 100 --warmup 10 --no-cpu-baseline
b bench_steps100_round_robin python bench.py --steps 100 --warmup 10 --no-cpu-baseline --threads 0
b bench_torchrun_1rank_rccl python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline
b bench_config3_swivel_chair_8192 python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline
b bench_config4_baxter_desk_mikael python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline
b bench_config5_mixed python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline
b bench_dense_reward python bench.py --dense --steps 100 --warmup 10 --no-cpu-baseline
b bench_osc_position_orientation python bench.py --control-type position_orientation --steps 100 --warmup 10 --no-cpu-baseline

#!/bin/bash
# The 1 / 2 / 4 / 8-GPU lines of the headline metric in one go, for whoever has the 8-GPU node (the builder's gpurun boxes have one GPU:
# RCCL has only ever run with one rank here -- profiles/r05_b_bench_torchrun_1rank_rccl.json).  Launch contract of the task statement:
# N > 1 runs under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1.  Output: one JSON line per N in $OUT (default
# gpurun_out/scale/), and the per-N value + scaling efficiency against N = 1 on stdout.
#   usage: scripts/scale_sweep.sh [STEPS=100] [WARMUP=10] [NS="1 2 4 8"]
cd "$(dirname "$0")/.." || exit 1
STEPS=${1:-100}; WARMUP=${2:-10}; NS=${3:-"1 2 4 8"}; OUT=${OUT:-gpurun_out/scale}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
have=$(python -c "import torch; print(torch.cuda.device_count())")
for n in $NS; do
  if [ "$n" -gt "$have" ]; then echo "N=$n: only $have GPU(s) visible -- skipped"; continue; fi
  f=$OUT/bench_n$n.json
  if [ "$n" -eq 1 ]; then python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" > "$f" 2> "$OUT/bench_n$n.err"
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARMUP" > "$f" 2> "$OUT/bench_n$n.err"; fi
  tail -1 "$f" > "$f.tmp" && mv "$f.tmp" "$f"
done
python - "$OUT" <<'PY'
import glob, json, os, sys
rows = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_n*.json"))):
    try:
        d = json.load(open(f)); rows[d["n_gpus"]] = d
    except Exception as e:
        print(f, "unreadable:", e)
if 1 in rows:
    v1 = rows[1]["value"]
    for n in sorted(rows):
        d = rows[n]
        print("N=%d  %10.0f env-steps/s  %.3f ms/step  weak-scaling efficiency %.3f  (rccl_world %s)" % (n, d["value"], d["ms_per_step"], d["value"] / (n * v1), d["config"].get("rccl_world")))
PY

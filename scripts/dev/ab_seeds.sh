#!/bin/bash
# bench A/B over several env/action seeds (a change that alters rounding alters WHICH envs grip: one seed cannot judge it):
#   ab_seeds.sh <outdir> <tag> ...   (tag -> furniture_amd/csrc/libfsim_<tag>.so; "base" = libfsim.so)
O=gpurun_out/$1; mkdir -p $O; shift
for seed in 1 2 3 4 5 6; do
for tag in "$@"; do
  lib=furniture_amd/csrc/libfsim_$tag.so; [ "$tag" = base ] && lib=furniture_amd/csrc/libfsim.so
  FSIM_BENCH_SEED=$seed FSIM_LIB=$PWD/$lib timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_${tag}_$seed.json 2> $O/bench_${tag}_$seed.err
done
done
python - $O "$@" <<'PY'
import json, sys, glob
O, tags = sys.argv[1], sys.argv[2:]
for t in tags:
    v = []
    for s in range(1, 7):
        try: v.append(json.load(open("%s/bench_%s_%d.json" % (O, t, s)))["value"])
        except Exception as e: pass
    print("SEEDS %-8s mean %.0f  per seed %s" % (t, sum(v) / max(1, len(v)), [round(x / 1e3) for x in v]))
PY

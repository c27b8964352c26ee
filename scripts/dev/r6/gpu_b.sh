#!/bin/bash
# round 6, second GPU call: (1) what removing the host turn-around between a slab's steps is worth (FSIM_EXP_AHEAD experiment build),
# (2) the Baxter / Cursor catalogue sweeps (discovery run of tests/test_agents_catalogue_gpu.py)
mkdir -p gpurun_out/r6b
for rep in 1 2; do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > gpurun_out/r6b/base_100_$rep.json 2> gpurun_out/r6b/base_100_$rep.err
  FSIM_EXP_AHEAD=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 > gpurun_out/r6b/ahead_100_$rep.json 2> gpurun_out/r6b/ahead_100_$rep.err
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --episode-window 0 --threads 1 > gpurun_out/r6b/threads_100_$rep.json 2> gpurun_out/r6b/threads_100_$rep.err
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --episode-window 0 > gpurun_out/r6b/base_20.json 2> gpurun_out/r6b/base_20.err
FSIM_EXP_AHEAD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --episode-window 0 > gpurun_out/r6b/ahead_20.json 2> gpurun_out/r6b/ahead_20.err
for f in gpurun_out/r6b/*.json; do echo $f $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']), d['ms_per_step'], d['roofline']['kernel_avg_ms'])"); done
( time timeout 1500 python -m pytest tests/test_agents_catalogue_gpu.py -q -s ) > gpurun_out/r6b/catalogue.txt 2>&1
grep -n "ran \|passed\|failed\|Error" gpurun_out/r6b/catalogue.txt | cut -c1-600 | head -30

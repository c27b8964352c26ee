#!/bin/bash
# round 6: soak of the committed library (3000 steps = 20 episodes per env: every reset from a look-ahead shadow, overflow re-steps counted), then the GPU suite
O=gpurun_out/r6i; mkdir -p $O
python bench.py --steps 3000 --warmup 100 --no-cpu-baseline --episode-window 0 > $O/bench_soak_3000.json 2> $O/bench_soak_3000.err
python -c "
import json; d=json.load(open('$O/bench_soak_3000.json')); c=d['config']
print('soak', round(d['value']), d['ms_per_step'], 'resets', c['resets_in_timed_region'], 'from lookahead', c['resets_taken_from_lookahead'], 'dropped', c['envs_that_dropped_contacts'], 'resteps', c['overflow_resteps'], 'finite', c['obs_finite'])"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python -c "
import json; d=json.load(open('$O/bench_driver.json')); print('driver', round(d['value']), d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['cpu_baseline']['value'])"
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt

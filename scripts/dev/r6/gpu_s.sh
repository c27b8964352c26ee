#!/bin/bash
# round 6, final library: the bench lines of the other configs again, a 3000-step soak, and the divergence control on the benchmark episode
bash scripts/dev/r6/bench_lines.sh r6s_lines
O=gpurun_out/r6s; mkdir -p $O
python bench.py --steps 3000 --warmup 100 --no-cpu-baseline --episode-window 0 > $O/bench_soak_3000.json 2> $O/bench_soak_3000.err
python -c "
import json; d=json.load(open('$O/bench_soak_3000.json')); c=d['config']
print('soak', round(d['value']), d['ms_per_step'], 'resets', c['resets_in_timed_region'], 'from lookahead', c['resets_taken_from_lookahead'], 'dropped', c['envs_that_dropped_contacts'], 'resteps', c['overflow_resteps'], 'finite', c['obs_finite'])"
make -C oracle -s
( time python scripts/divergence_control.py 1024 150 152 ) > $O/divergence_control.txt 2> $O/divergence_control.err; tail -4 $O/divergence_control.txt | cut -c1-300

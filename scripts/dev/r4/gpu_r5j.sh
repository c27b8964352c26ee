#!/bin/bash
O=gpurun_out/r5j; mkdir -p $O
export PYTHONPATH=$PWD
for t in test_overflow_restep_gpu test_determinism_gpu test_lookahead_gpu test_vec_env_gpu test_gpu_parity; do
  timeout 900 python -m pytest tests/$t.py -q -m gpu -x 2>&1 | tail -2 > $O/$t.txt; echo "$t: $(tail -1 $O/$t.txt)"
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/err.txt; python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(round(d['value']))"

#!/bin/bash
# round 6: the GPU suite in its variant modes (every env on a four-wave team; every CU's LDS NaN-filled before every launch; generic kernels),
# the new second-rung test, the divergence control for the Cursor agent
O=gpurun_out/r6g; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_overflow_restep_gpu.py -q -x ) > $O/pytest_restep.txt 2>&1; tail -3 $O/pytest_restep.txt
( time python scripts/divergence_control.py 256 30 62 Cursor toy_table ) > $O/divergence_control_cursor_toy_table.txt 2> $O/divergence_control_cursor.err; tail -4 $O/divergence_control_cursor_toy_table.txt | cut -c1-300
( time FSIM_MW_K=0 timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_mwk0.txt 2>&1; tail -3 $O/pytest_mwk0.txt
( time FSIM_TEST_POISON=7fc00000 timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_poison.txt 2>&1; tail -3 $O/pytest_poison.txt
( time FSIM_GENERIC=1 timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_generic.txt 2>&1; tail -3 $O/pytest_generic.txt

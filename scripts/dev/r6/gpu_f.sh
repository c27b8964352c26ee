#!/bin/bash
# round 6: the Cursor agent, device against the native checker; the catalogue sweeps with their printed lists
O=gpurun_out/r6f; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_capi_cpu.py -m gpu -x -q -k "cursor" -s ) > $O/pytest_cursor.txt 2>&1
tail -15 $O/pytest_cursor.txt


grep -n "envs within\|passed\|failed" $O/pytest_cursor.txt | cut -c1-900

#!/bin/bash
O=gpurun_out/r6j; mkdir -p $O
( time timeout 1800 python -m pytest tests/test_capi_cpu.py -m gpu -q -s -k "other_agents" ) > $O/pytest_other_agents.txt 2>&1
grep -n "envs within\|passed\|failed\|Error\|assert " $O/pytest_other_agents.txt | cut -c1-300 | head -40

#!/bin/bash
# round 6: the dense reward device vs native; the whole GPU suite as the driver runs it
O=gpurun_out/r6h; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_capi_cpu.py -m gpu -x -q -k "dense" ) > $O/pytest_dense.txt 2>&1; tail -12 $O/pytest_dense.txt | cut -c1-400
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

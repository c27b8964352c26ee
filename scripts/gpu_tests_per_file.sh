#!/bin/bash
# every GPU test file in a process of its own (a GPU abort in one file must not lose the other files' reports); summary at the end
O=${1:-gpurun_out/gpu_tests}; mkdir -p $O
export PYTHONPATH=$PWD
rc=0
for f in tests/test_*.py; do
  grep -q "mark.gpu\|pytestmark = pytest.mark.gpu" $f || continue
  n=$(basename $f .py)
  timeout 1500 python -m pytest $f -q -m gpu -x 2>&1 | tail -25 > $O/$n.txt
  line=$(grep -E "passed|failed|error|no tests ran" $O/$n.txt | tail -1)
  echo "$n: $line"
  echo "$n: $line" >> $O/summary.txt
  echo "$line" | grep -qE "failed|error" && rc=1
done
exit $rc

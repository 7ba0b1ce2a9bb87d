#!/bin/bash
# How often does tests/test_hip_c5.py::test_c5_batch_on_one_gpu_matches_oracle[1-16] fail?  usage: bash tools/flaky_c5.sh DIR RUNS
D=${1:-.}; RUNS=${2:-20}
cd $GRAFT_REPO_ROOT/$D
f=0
for i in $(seq $RUNS); do
  timeout 120 python -m pytest tests/test_hip_c5.py -x -q -m gpu -k "test_c5_batch_on_one_gpu_matches_oracle and 1-16" > /tmp/flaky.log 2>&1 || { f=$((f+1)); grep -E "^E +(assert|Assertion)" /tmp/flaky.log | head -3; grep -E "Error" /tmp/flaky.log | head -2; }
done
echo "$D: $f failures of $RUNS"

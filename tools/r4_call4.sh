#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c4; rm -rf $O; mkdir -p $O
timeout 300 bash tools/fact_tail_trace.sh > $O/tail_gated.txt 2>&1
MNK_OPTIONS="dag_gated_inv=0" timeout 300 bash tools/fact_tail_trace.sh > $O/tail_ungated.txt 2>&1
for i in 1 2 3; do timeout 300 python -m pytest tests/test_hip_round4.py -q -k "take_turns or kkt_handle" 2>&1 | grep -v "^$" | tail -12 >> $O/t_threads.log; done
cat $O/tail_gated.txt | tail -45; echo; tail -30 $O/tail_ungated.txt; cat $O/t_threads.log | grep -n "passed\|failed\|AssertionError" 

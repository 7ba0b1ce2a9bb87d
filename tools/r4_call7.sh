#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c7; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
MNK_OPTIONS="" timeout 120 python tools/thread_stress.py 4100 6 8 2>&1 | tail -1 >> $O/stress.txt
MNK_OPTIONS="dag_chain_inline=0" timeout 120 python tools/thread_stress.py 4100 6 8 2>&1 | tail -1 >> $O/stress.txt
MNK_OPTIONS="" timeout 120 python tools/thread_stress.py 7000 6 6 2>&1 | tail -1 >> $O/stress.txt
done
cat $O/stress.txt

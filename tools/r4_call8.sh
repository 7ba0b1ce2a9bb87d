#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c8; rm -rf $O; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
timeout 120 python tools/thread_stress.py 7000 6 6 2>&1 | tail -1 >> $O/stress.txt
done
for i in 1 2 3 4; do
timeout 120 python tools/thread_stress.py 4100 6 8 2>&1 | tail -1 >> $O/stress.txt
done
cat $O/stress.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c13; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for d in _r3ab . _vA _vB; do
  (cd $d; echo "[$d] $(MNK_OPTIONS=dag_fill=0 timeout 100 python tools/dag_time.py 11192 LDL 2>/dev/null | tail -1)" >> $GRAFT_REPO_ROOT/$O/t.txt)
done
done
cat $O/t.txt

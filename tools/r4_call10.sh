#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c10; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
for d in _r3ab .; do
  (cd $d; echo "[$d] $(timeout 100 python tools/dag_time.py 11192 LDL 2>/dev/null | tail -1)" >> $GRAFT_REPO_ROOT/$O/dagtime.txt)
done
done
(cd _r3ab; timeout 100 python tools/dag_chain.py 11192 LDL > $GRAFT_REPO_ROOT/$O/chain_r3.txt 2>&1)
(timeout 100 python tools/dag_chain.py 11192 LDL > $O/chain_r4.txt 2>&1)
(cd _r3ab; timeout 100 python tools/dag_util.py 11192 LDL > $GRAFT_REPO_ROOT/$O/util_r3.txt 2>&1)
(timeout 100 python tools/dag_util.py 11192 LDL > $O/util_r4.txt 2>&1)
cat $O/dagtime.txt; head -30 $O/chain_r4.txt

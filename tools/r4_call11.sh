#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c11; rm -rf $O; mkdir -p $O
for d in _r3ab .; do
  (cd $d; echo "[$d]" >> $GRAFT_REPO_ROOT/$O/ptrs.txt; MNK_DEBUG_PTRS=1 timeout 100 python tools/dag_time.py 11192 LDL 2>&1 | grep "PTRS\|factorize" | sort | uniq -c >> $GRAFT_REPO_ROOT/$O/ptrs.txt)
done
for d in _r3ab .; do
  (cd $d; echo "[$d bench]" >> $GRAFT_REPO_ROOT/$O/ptrs.txt; MNK_DEBUG_PTRS=1 timeout 100 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-c4 --no-ipm-loop 2>&1 | grep "PTRS" | sort | uniq -c >> $GRAFT_REPO_ROOT/$O/ptrs.txt)
done
for v in "" "dag_fill=0"; do echo "[r4 $v] $(MNK_OPTIONS=$v timeout 100 python tools/dag_time.py 11192 LDL 2>/dev/null | tail -1)" >> $O/ptrs.txt; done
cat $O/ptrs.txt

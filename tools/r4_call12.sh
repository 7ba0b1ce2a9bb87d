#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c12; rm -rf $O; mkdir -p $O
(cd _r3ab; echo "[r3] $(timeout 100 python tools/dag_time.py 11192 LDL 2>/dev/null | tail -1)" >> $GRAFT_REPO_ROOT/$O/t.txt)
echo "[r4] $(timeout 100 python tools/dag_time.py 11192 LDL 2>/dev/null | tail -1)" >> $O/t.txt
echo "[r4 nobatchstreams] $(MNK_NO_BATCH_STREAMS=1 timeout 100 python tools/dag_time.py 11192 LDL 2>/dev/null | tail -1)" >> $O/t.txt
(cd _r3ab; echo "[r3] $(timeout 100 python tools/dag_time.py 11192 LDL 2>/dev/null | tail -1)" >> $GRAFT_REPO_ROOT/$O/t.txt)
echo "[r4 nobatchstreams] $(MNK_NO_BATCH_STREAMS=1 timeout 100 python tools/dag_time.py 11192 LDL 2>/dev/null | tail -1)" >> $O/t.txt
echo "[r4] $(timeout 100 python tools/dag_time.py 11192 LDL 2>/dev/null | tail -1)" >> $O/t.txt
(cd _r3ab; timeout 100 python tools/dag_util.py 11192 LDL > $GRAFT_REPO_ROOT/$O/util_r3.txt 2>&1)
(MNK_OPTIONS=dag_fill=0 timeout 100 python tools/dag_util.py 11192 LDL > $O/util_r4.txt 2>&1)
cat $O/t.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c19; rm -rf $O; mkdir -p $O
export ROCP_TOOL_LIBRARIES=$GRAFT_REPO_ROOT/tools/devcount/libmnk_devcount.so
for s in mfma fetch write; do
  timeout 200 python tools/devcount_dag.py $s 20 > $O/dc_$s.json 2> $O/dc_$s.err
  tail -c 900 $O/dc_$s.json; echo; tail -3 $O/dc_$s.err
done
unset ROCP_TOOL_LIBRARIES
timeout 300 python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline 2>> $O/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[batch16]', round(d['value'],2), round(d['ms_per_step'],3), round(d['ms_per_factorize'],3), round(d['ms_per_solve'],3))"

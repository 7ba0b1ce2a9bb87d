#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c3; rm -rf $O; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-ipm-loop"
for v in "" "dag_fill=0" "dag_gated_inv=0" "dag_fill=0,dag_gated_inv=0" ""; do
  MNK_OPTIONS="$v" timeout 200 $B 2>> $O/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', round(d['value'],2), round(d['ms_per_step'],3), round(d['ms_per_factorize'],3), round(d['ms_per_solve'],3), d['roofline']['schedule_panel_algo'], d['roofline']['pp_fallbacks'])" >> $O/ab.txt
done
timeout 600 python -m pytest tests/test_hip_round4.py tests/test_hip_c5.py -q 2>&1 | tail -30 > $O/t.log
cat $O/ab.txt; tail -5 $O/t.log

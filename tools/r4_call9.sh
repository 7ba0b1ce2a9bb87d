#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c9; rm -rf $O; mkdir -p $O
B="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-ipm-loop"
for rep in 1 2; do
for d in _r3ab .; do
  (cd $d; timeout 200 python $B 2>> $GRAFT_REPO_ROOT/$O/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$d]', round(d['value'],2), round(d['ms_per_step'],3), round(d['ms_per_factorize'],3), round(d['ms_per_solve'],3), round(d['ms_assemble'],4), d['roofline']['schedule_panel_algo'], d['roofline']['pp_fallbacks'])" >> $GRAFT_REPO_ROOT/$O/ab.txt)
done
done
for a in "--batch 16" "--batch 16 --no-batch-api" "--batch 16 --concurrency 4"; do
timeout 300 python bench.py --steps 5 --warmup 2 $a --no-cpu-baseline 2>> $O/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$a]', round(d['value'],2), round(d['ms_per_step'],3), round(d['ms_per_factorize'],3), round(d['ms_per_solve'],3), d['roofline']['schedule_panel_algo'], d['roofline']['pp_fallbacks'])" >> $O/ab.txt
done
for i in 1 2; do timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 >> $O/t_full.log; done
cat $O/ab.txt; grep -n "passed\|failed\|Assertion\|^E  " $O/t_full.log

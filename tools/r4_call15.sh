#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c16; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_c5.py -q -x 2>&1 | tail -25 > $O/t_c5.log
for a in "--batch 16" "--batch 16 --no-batch-api" "--batch 16 --concurrency 4" "--batch 4"; do
timeout 300 python bench.py --steps 5 --warmup 2 $a --no-cpu-baseline 2>> $O/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$a]', round(d['value'],2), round(d['ms_per_step'],3), round(d['ms_per_factorize'],3), round(d['ms_per_solve'],3), d['roofline']['schedule_panel_algo'], d['roofline']['pp_fallbacks'])" >> $O/ab.txt
done
for q in 2 8; do
MNK_SOLVE_BATCH_Q=$q timeout 300 python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline 2>> $O/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[Q=$q]', round(d['value'],2), round(d['ms_per_step'],3))" >> $O/ab.txt
done
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-ipm-loop 2>> $O/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[C3]', round(d['value'],2), round(d['ms_per_step'],3), round(d['ms_per_factorize'],3), round(d['ms_per_solve'],3))" >> $O/ab.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 >> $O/t_full.log
tail -8 $O/t_c5.log; cat $O/ab.txt; grep -n "passed\|failed\|Assertion\|^E  " $O/t_full.log; tail -3 $O/err.log

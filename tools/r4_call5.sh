#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c5; rm -rf $O; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_hip_round4.py -q -k "take_turns" 2>&1 | grep -v "^$" | tail -4 >> $O/t_threads.log; done
timeout 900 python -m pytest tests -m gpu -q 2>&1 > $O/t_full_all.log; tail -30 $O/t_full_all.log > $O/t_full.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-ipm-loop"
for v in "" "dag_fill=0" ""; do
  MNK_OPTIONS="$v" timeout 200 $B 2>> $O/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', round(d['value'],2), round(d['ms_per_step'],3), round(d['ms_per_factorize'],3), round(d['ms_per_solve'],3), d['roofline']['schedule_panel_algo'], d['roofline']['pp_fallbacks'])" >> $O/ab.txt
done
for a in "--batch 16" "--batch 16 --no-batch-api" "--batch 16 --concurrency 4" "--batch 8" "--batch 4" "--batch 2"; do
timeout 300 python bench.py --steps 5 --warmup 2 $a --no-cpu-baseline 2>> $O/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$a]', round(d['value'],2), round(d['ms_per_step'],3), round(d['ms_per_factorize'],3), round(d['ms_per_solve'],3), d['roofline']['schedule_panel_algo'], d['roofline']['pp_fallbacks'])" >> $O/ab.txt
done
grep -c passed $O/t_threads.log; grep -n "failed\|Assertion" $O/t_threads.log | head; tail -6 $O/t_full.log; cat $O/ab.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c18; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_c5.py -q -x 2>&1 | tail -5 > $O/t_c5.log
for a in "--batch 16" "--batch 16 --concurrency 4" "--batch 16 --no-batch-api" "--batch 16"; do
timeout 300 python bench.py --steps 5 --warmup 2 $a --no-cpu-baseline 2>> $O/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$a]', round(d['value'],2), round(d['ms_per_step'],3), round(d['ms_per_factorize'],3), round(d['ms_per_solve'],3), d['roofline']['schedule_panel_algo'], d['roofline']['pp_fallbacks'])" >> $O/ab.txt
done
bash tools/batch_trace.sh 2>&1 | grep "step span\|chains\|solves" >> $O/ab.txt
tail -3 $O/t_c5.log; cat $O/ab.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c17; rm -rf $O; mkdir -p $O
run() { # label, env...
  lab="$1"; shift
  env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline 2>> $O/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$lab]', round(d['value'],2), round(d['ms_per_step'],3), round(d['ms_per_factorize'],3), d['roofline']['schedule_panel_algo'], d['roofline']['pp_fallbacks'])" >> $O/ab.txt
}
run default X=1
run band8 MNK_DAG_CUS=8 MNK_OPTIONS=dag_band=8
run band12 MNK_DAG_CUS=12 MNK_OPTIONS=dag_band=12
run period48 MNK_OPTIONS=batch_period=48
run period56 MNK_OPTIONS=batch_period=56
run period66 MNK_OPTIONS=batch_period=66
run default X=1
cat $O/ab.txt; tail -3 $O/err.log

#!/bin/bash
# A/B of the deferred tail-region schedule (MNK_DEFER_ROWS) on the bench workload
mkdir -p gpurun_out
run() { timeout 120 python bench.py --no-cpu-baseline --no-ipm-loop --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('  it/s %.2f  factorize %.3f ms  solve %.3f' % (j['value'], j['ms_per_factorize'], j['ms_per_solve']))"; }
for d in 0 2048 3072 4096 5120 6144; do
  for sm in 1 0; do
  echo "== MNK_DEFER_ROWS=$d small=$sm"; MNK_DEFER_ROWS=$d MNK_DEFER_SMALL=$sm run
  done
done

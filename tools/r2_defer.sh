#!/bin/bash
# A/B of the deferred tail-region schedule (MNK_DEFER_ROWS x MNK_DEFER_SPLIT) on the bench workload
mkdir -p gpurun_out
run() { timeout 120 python bench.py --no-cpu-baseline --no-ipm-loop --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('  it/s %.2f  factorize %.3f ms  solve %.3f' % (j['value'], j['ms_per_factorize'], j['ms_per_solve']))"; }
echo "== default"; run
for d in ${ROWS:-2048 3072 4096 5120}; do
  for sp in ${SPLITS:-4 8}; do
  echo "== MNK_DEFER_ROWS=$d split=$sp"; MNK_DEFER_ROWS=$d MNK_DEFER_SPLIT=$sp run
  done
done

#!/bin/bash
# generic A/B on the bench workload: tools/r2_ab.sh VAR v1 v2 ... (alternating, two rounds)
var=$1; shift
run() { timeout 120 python bench.py --no-cpu-baseline --no-ipm-loop --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('  it/s %.2f  factorize %.3f ms  solve %.3f' % (j['value'], j['ms_per_factorize'], j['ms_per_solve']))"; }
for rep in 1 2; do for v in "$@"; do echo "== $var=$v"; env $var=$v bash -c "$(declare -f run); run"; done; done

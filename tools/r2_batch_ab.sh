#!/bin/bash
for rep in 1 2 3; do for nb in 0 1; do
echo "== NO_BATCH=$nb"; MNK_IPM_NO_BATCH=$nb timeout 200 python tools/ipm_run_device.py $@ 2>/dev/null | grep device-resident | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    if j['run']==1: print('   ms/iter %.3f' % j['ms_per_iteration_wall'])"
done; done

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c2; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_c5.py -x -q 2>&1 | tail -40 > $O/t_c5.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 > $O/t_full_all.log; tail -60 $O/t_full_all.log > $O/t_full.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-ipm-loop 2> $O/bench1.err | grep '^{' > $O/bench1.json
timeout 300 python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline 2> $O/c5_b.err | grep '^{' > $O/c5_batchapi.json
timeout 300 python bench.py --steps 5 --warmup 2 --batch 16 --no-batch-api --no-cpu-baseline 2> $O/c5_nb.err | grep '^{' > $O/c5_nobatchapi.json
timeout 300 python bench.py --steps 5 --warmup 2 --batch 16 --concurrency 4 --no-cpu-baseline 2> $O/c5_b4.err | grep '^{' > $O/c5_batchapi_conc4.json
tail -8 $O/t_c5.log; tail -8 $O/t_full.log
for f in bench1 c5_batchapi c5_nobatchapi c5_batchapi_conc4; do python - <<PY
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],2), round(d["ms_per_step"],3), round(d["ms_per_factorize"],3), round(d["ms_per_solve"],3), d["roofline"]["schedule_panel_algo"], d["roofline"]["pp_fallbacks"])
except Exception as e: print("$f", "ERR", e)
PY
done
tail -3 $O/*.err

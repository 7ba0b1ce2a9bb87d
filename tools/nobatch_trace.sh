#!/bin/bash
# Kernel trace of bench.py --batch 16 --no-batch-api (16 instances back to back on one context, no batch API)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/nobatch
rm -rf $R; mkdir -p $R
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/t -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --batch 16 --no-batch-api --steps 2 --warmup 1 > $R/run.log 2>&1
grep '^{' $R/run.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $R/t -name "*.db" | head -1) $R/stats.md | head -14
python tools/trace_dump.py $(find $R/t -name "*.db" | head -1) $R/trace.csv 2>/dev/null
rm -rf $R/t
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/trace.csv")))
rows.sort(key=lambda r:int(r['start']))
pc=[r for r in rows if 'pchain' in r['name']]
last=pc[-20:]
t0=int(last[0]['start'])
print("last chains (start, duration ms):", " ".join(f"[{(int(r['start'])-t0)/1e6:.1f} +{(int(r['end'])-int(r['start']))/1e6:.1f}]" for r in last))
PY

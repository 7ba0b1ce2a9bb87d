#!/bin/bash
# kernel timeline of the last factorize! of config C2 (tools/bench_configs.py c2) under rocprofv3 --kernel-trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/c2trace
rm -rf $R; mkdir -p $R
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $R/t -o p -- python $GRAFT_REPO_ROOT/tools/bench_configs.py c2 > $R/run.log 2>&1
grep '^{' $R/run.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python tools/trace_dump.py $(find $R/t -name "*.db" | head -1) $R/trace.csv 2>/dev/null
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/trace.csv")))
idx=[i for i,r in enumerate(rows) if 'copy_lower' in r['name']]
# the 4th from last copy_lower: a timed factorization of the second config; print until the next copy_lower
i0=idx[-3]; i1=idx[-2]
sel=rows[i0:i1]
t0=int(sel[0]['start'])
for r in sel:
    s=(int(r['start'])-t0)/1e3; e=(int(r['end'])-t0)/1e3
    print(f"{r['name'][:44]:44s} start {s:8.1f} end {e:8.1f} dur {e-s:7.1f}")
PY

#!/bin/bash
# kernels of the last factorize! of bench.py around the end of the pivot chain (rocprofv3 --kernel-trace)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/tailtrace
rm -rf $R; mkdir -p $R
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $R/t -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-ipm-loop --no-c4 --steps 4 --warmup 1 > $R/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_dump.py $(find $R/t -name "*.db" | head -1) $R/trace.csv 2>/dev/null
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/trace.csv")))
idx=[i for i,r in enumerate(rows) if 'scatter_csc' in r['name']]
i0=idx[-1]
sel=rows[i0:i0+40]
pc=[r for r in sel if 'pchain' in r['name']][0]
tend=int(pc['end'])
for r in sel:
    s=(int(r['start'])-tend)/1e3; e=(int(r['end'])-tend)/1e3
    if e > -600: print(f"{r['name'][:48]:48s} start {s:8.1f} end {e:8.1f} dur {e-s:7.1f}")
    if 'persistent_solve' in r['name']: break
PY

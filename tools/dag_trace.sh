#!/bin/bash
# Kernel timeline of the last factorization of tools/dag_check.py under rocprofv3 (GPU box). usage: tools/dag_trace.sh N ALGS(LDL|CHOLESKY) [algo]
export TMPDIR=/tmp
N=${1:-11192}; ALG=${2:-LDL}; PA=${3:-5}
R=$GRAFT_REPO_ROOT/gpurun_out/dagtrace
rm -rf $R; mkdir -p $R
cd /tmp
DAG_ONLY=$ALG DAG_ALGOS=$PA DAG_REPS=3 timeout 200 rocprofv3 --kernel-trace -d $R/t -o p -- python $GRAFT_REPO_ROOT/tools/dag_check.py $N > $R/run.log 2>&1
tail -3 $R/run.log
cd $GRAFT_REPO_ROOT
python tools/trace_dump.py $(find $R/t -name "*.db" | head -1) $R/trace.csv 2>/dev/null
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/trace.csv")))
# last factorization: from the last fill_lower/copy_lower kernel on
idx=[i for i,r in enumerate(rows) if 'fill_lower' in r['name'] or 'copy_lower' in r['name']]
i0=idx[-1]
sel=rows[i0:]
# cut at the first solve kernel
end=len(sel)
for k,r in enumerate(sel):
    if 'solve' in r['name'] and k>3: end=k; break
sel=sel[:end]
t0=int(sel[0]['start'])
print("n kernels", len(sel), "span us", (max(int(r['end']) for r in sel)-t0)/1e3)
import os
if os.environ.get("DAG_TRACE_FULL"):
    for r in sel:
        s=(int(r['start'])-t0)/1e3; e=(int(r['end'])-t0)/1e3
        print(f"{r['name'][:40]:40s} start {s:9.1f} dur {e-s:8.1f} grid {r.get('grid_x', r.get('grid_size_x','?'))}")
pp=[r for r in sel if 'ppanel' in r['name']]
print("ppanel starts:", " ".join(f"{(int(r['start'])-t0)/1e3:.0f}" for r in pp))
print("ppanel durs  :", " ".join(f"{(int(r['end'])-int(r['start']))/1e3:.0f}" for r in pp))
for r in sel:
    if 'bulk' in r['name']: print("bulk start", (int(r['start'])-t0)/1e3, "dur", (int(r['end'])-int(r['start']))/1e3)
PY

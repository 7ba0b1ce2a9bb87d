#!/bin/bash
# Kernel timeline of one batched step (bench.py --batch 16): the merged bulk launch, the two chains' launches, inverses, solves.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/batchtrace
rm -rf $R; mkdir -p $R
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/t -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --batch 16 --steps 3 --warmup 1 > $R/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_dump.py $(find $R/t -name "*.db" | head -1) $R/trace.csv 2>/dev/null
rm -rf $R/t
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/trace.csv")))
rows.sort(key=lambda r:int(r['start']))
bulk=[i for i,r in enumerate(rows) if 'dag_bulk_kernel<' in r['name'] and int(r['end'])-int(r['start'])>20e6]
print("merged bulk launches:", len(bulk))
i=bulk[-2]
t0=int(rows[i]['start']); t1=int(rows[bulk[-1]]['start'])
print(f"step span (bulk start to next bulk start): {(t1-t0)/1e6:.2f} ms; bulk kernel {(int(rows[i]['end'])-t0)/1e6:.2f} ms")
# per-kernel-name totals within the step window
import collections
tot=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    s=int(r['start'])
    if t0-2e6 <= s < t1-2e6:
        n=r['name'][:40]; tot[n][0]+=1; tot[n][1]+=(int(r['end'])-s)/1e6
for n,(c,ms) in sorted(tot.items(), key=lambda kv:-kv[1][1])[:18]: print(f"  {n:42s} x{c:4d} {ms:9.3f} ms")
# chains: start/end of each pchain
pc=[r for r in rows if 'pchain' in r['name'] and t0-1e6 <= int(r['start']) < t1-2e6]
print("chains:", " ".join(f"[{(int(r['start'])-t0)/1e6:.1f}-{(int(r['end'])-t0)/1e6:.1f}]" for r in pc))
ps=[r for r in rows if 'persistent_solve' in r['name'] and t0 <= int(r['start']) < t1+40e6]
if ps: print(f"solves: first start {(int(ps[0]['start'])-t0)/1e6:.2f} ms, last end {(int(ps[-1]['end'])-t0)/1e6:.2f} ms, n={len(ps)}, mean {sum(int(r['end'])-int(r['start']) for r in ps)/len(ps)/1e3:.1f} us")
PY

#!/bin/bash
# Kernel timeline of tools/thread_stress.py around a device-side time-out (the longest pchain_kernel of the run): what ran
# on the device while the chain waited.  usage: bash tools/stress_trace.sh [N T R]  -> gpurun_out/stresstrace
export TMPDIR=/tmp
N=${1:-11192}; T=${2:-4}; RR=${3:-3}
R=$GRAFT_REPO_ROOT/gpurun_out/stresstrace
rm -rf $R; mkdir -p $R
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/t -o p -- python $GRAFT_REPO_ROOT/tools/thread_stress.py $N $T $RR > $R/run.log 2>&1
tail -1 $R/run.log
cd $GRAFT_REPO_ROOT
python tools/trace_dump.py $(find $R/t -name "*.db" | head -1) $R/trace.csv 2>/dev/null
rm -rf $R/t
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/trace.csv")))
rows.sort(key=lambda r:int(r['start']))
pc=[r for r in rows if 'pchain' in r['name']]
worst=max(pc, key=lambda r:int(r['end'])-int(r['start']))
t0,t1=int(worst['start']),int(worst['end'])
print(f"longest chain: {(t1-t0)/1e6:.1f} ms, columns: {list(worst.keys())}")
print("kernels that overlap it or start within 3 ms before / 1 ms after (start, end relative to the chain's start, ms):")
for r in rows:
    s,e=int(r['start']),int(r['end'])
    if e >= t0-3e6 and s <= t1+1e6:
        extra=" ".join(f"{k}={r[k]}" for k in r if k in ('stream','queue','stream_id','queue_id','tid','thread_id'))
        print(f"  {(s-t0)/1e6:10.3f} {(e-t0)/1e6:10.3f}  {r['name'][:60]:60s} {extra}")
PY

#!/bin/bash
# Kernel timeline of ONE bench step (rocprofv3 --kernel-trace of bench.py): every kernel between the first kernel of a
# step's assembly and the first kernel of the next step's, with the idle gaps -- where the time outside the three phases
# (factorize! / solve! / assembly) goes.  -> gpurun_out/stepgaps/step.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/stepgaps
rm -rf $R; mkdir -p $R
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/t -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-ipm-loop --no-c4 --steps 6 --warmup 2 > $R/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_dump.py $(find $R/t -name "*.db" | head -1) $R/trace.csv 2>/dev/null
python - <<PY > $R/step.txt
import csv
rows=list(csv.DictReader(open("$R/trace.csv")))
rows.sort(key=lambda r:int(r['start']))
idx=[i for i,r in enumerate(rows) if 'segsum' in r['name']]
# steps start with two segsum launches (compress_jacobian, compress_hessian): take the first of each pair
starts=[i for k,i in enumerate(idx) if k==0 or idx[k]-idx[k-1]>2]
if len(starts)>=5:
    i0,i1=starts[-4],starts[-3]
    t0=int(rows[i0]['start'])
    prev_end=t0
    print(f"step: {(int(rows[i1]['start'])-t0)/1e3:.1f} us from first kernel to the next step's first kernel")
    busy=0
    for r in rows[i0:i1]:
        s=(int(r['start'])-t0)/1e3; e=(int(r['end'])-t0)/1e3
        gap=(int(r['start'])-prev_end)/1e3
        print(f"{r['name'][:60]:60s} start {s:9.1f} dur {e-s:8.1f} gap_before {gap:7.1f}")
        prev_end=max(prev_end,int(r['end']))
PY
rm -rf $R/t
tail -40 $R/step.txt

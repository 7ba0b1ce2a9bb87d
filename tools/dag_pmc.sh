#!/bin/bash
# PMC passes (FETCH_SIZE, TCC hit/miss, MFMA busy) of tools/dag_check.py on the GPU box. usage: tools/dag_pmc.sh N ALG algo
export TMPDIR=/tmp
N=${1:-11192}; ALG=${2:-LDL}; PA=${3:-5}
R=$GRAFT_REPO_ROOT/gpurun_out/dagpmc
rm -rf $R; mkdir -p $R
cd /tmp
run() { DAG_ONLY=$ALG DAG_ALGOS=$PA DAG_REPS=2 timeout 200 rocprofv3 --kernel-trace --pmc $2 -d $R/$1 -o p -- python $GRAFT_REPO_ROOT/tools/dag_check.py $N > $R/$1.log 2>&1; }
run fetch FETCH_SIZE
run hit "TCC_HIT_sum TCC_MISS_sum"
run mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
cd $GRAFT_REPO_ROOT
python - <<PY
import sqlite3, glob, re
def agg(d):
    f=glob.glob("$R/%s/**/*.db"%d, recursive=True)
    if not f: print(d,"no db"); return {}
    db=sqlite3.connect(f[0])
    rows=db.execute("select dispatch_id, name, start, end, counter_name, counter_value from pmc_events order by start").fetchall()
    disp={}
    for i,n,s,e,c,v in rows:
        r=disp.setdefault(i,{"name":n,"start":s,"end":e,"c":{}}); r["c"][c]=r["c"].get(c,0)+v
    order=sorted(disp.values(), key=lambda r:r["start"])
    st=[i for i,r in enumerate(order) if "fill_lower" in r["name"] or "copy_lower_kernel" in r["name"]]
    seg=order[st[-1]:]
    out={}
    for r in seg:
        n=re.sub(r"\(.*","",r["name"]).replace("void ","").replace("mnk::","")[:40]
        if "solve" in n: break
        a=out.setdefault(n,{"n":0,"t":0.0})
        a["n"]+=1; a["t"]+=(r["end"]-r["start"])/1e6
        for c,v in r["c"].items(): a[c]=a.get(c,0)+v
    return out
for d in ("fetch","hit","mfma"):
    o=agg(d)
    for k,v in sorted(o.items(), key=lambda kv:-kv[1]["t"])[:6]:
        extra=""
        if "FETCH_SIZE" in v: extra=f"read GB (x2) {2*v['FETCH_SIZE']*1024/1e9:.2f}"
        if "TCC_HIT_sum" in v: extra=f"L2 hit rate {v['TCC_HIT_sum']/(v['TCC_HIT_sum']+v['TCC_MISS_sum']+1e-9):.3f} hits {v['TCC_HIT_sum']:.3e} miss {v['TCC_MISS_sum']:.3e}"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v: extra=f"MFMA busy {v['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*v['GRBM_GUI_ACTIVE']/8+1e-9):.3f}"
        print(f"{d:6s} {k:40s} n={v['n']:3d} t={v['t']:8.2f} ms {extra}")
PY

#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2f
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $R/pytest.log
tail -12 $R/pytest.log
for algo in 3 1; do
  MNK_PANEL_ALGO=$algo timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/bench_algo$algo.log 2>&1
  python - <<PY
import json
l=[x for x in open("$R/bench_algo$algo.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("algo$algo", "factorize", d["ms_per_factorize"], "solve", d["ms_per_solve"], "it/s", d["value"], "frac", d["roofline"]["frac"])
else: print(open("$R/bench_algo$algo.log").read()[-500:])
PY
done
MNK_PANEL_ALGO=3 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --algorithm CHOLESKY 2>&1 | tail -1 | cut -c1-200
timeout 200 python tools/bench_configs.py c2 2>&1 | cut -c1-260
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/trace -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/trace_bench.log 2>&1
DB=$(ls $R/trace/*/*.db 2>/dev/null | head -1); [ -z "$DB" ] && DB=$(ls $R/trace/*.db | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB $R/kernel_stats.md | head -16
python $GRAFT_REPO_ROOT/tools/trace_dump.py $DB $R/trace.csv 6000 2>/dev/null
rm -rf $R/trace

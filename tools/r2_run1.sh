#!/bin/bash
# round-2 GPU pass 1: tests, A/B of the panel step, kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2b
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $R/pytest.log
tail -4 $R/pytest.log
for algo in 1 0; do
  MNK_PANEL_ALGO=$algo timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/bench_algo$algo.log 2>&1
  python - <<PY
import json
l=[x for x in open("$R/bench_algo$algo.log") if x.startswith("{")]
d=json.loads(l[-1]); print("algo$algo", "factorize", d["ms_per_factorize"], "solve", d["ms_per_solve"], "it/s", d["value"], "frac", d["roofline"]["frac"])
PY
done
MNK_PANEL_ALGO=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --algorithm CHOLESKY > $R/bench_chol.log 2>&1
tail -c 600 $R/bench_chol.log | head -c 400; echo
timeout 200 python tools/bench_configs.py c2 > $R/c2.log 2>&1; cat $R/c2.log | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/trace -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/trace_bench.log 2>&1
DB=$(ls $R/trace/*/*.db 2>/dev/null | head -1); [ -z "$DB" ] && DB=$(ls $R/trace/*.db | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB $R/kernel_stats.md | head -40
rm -rf $R/trace

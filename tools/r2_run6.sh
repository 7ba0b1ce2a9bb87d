#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2i
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "cholesky_vs_lapack or ldl_inertia or schedules_agree or deterministic or known_answer" 2>&1 | tail -6
for ov in 1 0; do
  MNK_OVERLAP=$ov timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/bench_ov$ov.log 2>&1
  python - <<PY
import json
l=[x for x in open("$R/bench_ov$ov.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("overlap=$ov", "factorize", d["ms_per_factorize"], "solve", d["ms_per_solve"], "it/s", d["value"], "frac", d["roofline"]["frac"])
else: print(open("$R/bench_ov$ov.log").read()[-800:])
PY
done
timeout 200 python tools/bench_configs.py c2 2>&1 | cut -c1-260
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6

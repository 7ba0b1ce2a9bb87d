#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2e
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $R/pytest.log
tail -25 $R/pytest.log
for p0 in 1 0; do
  MNK_PANEL0_WHOLE=$p0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/bench_p0_$p0.log 2>&1
  python - <<PY
import json
l=[x for x in open("$R/bench_p0_$p0.log") if x.startswith("{")]
d=json.loads(l[-1]); print("panel0_whole=$p0", "factorize", d["ms_per_factorize"], "solve", d["ms_per_solve"], "it/s", d["value"], "frac", d["roofline"]["frac"])
PY
done
timeout 200 python tools/bench_configs.py c2 > $R/c2.log 2>&1; cat $R/c2.log | cut -c1-300

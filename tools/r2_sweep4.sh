#!/bin/bash
# persistent panel kernel: second sweep (acquire batching in; NB = 8 launches for the tail; width of the panel stream's own (a) piece)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2f
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
run() { # label, env..., -- args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > $R/$label.log 2>&1
  python - <<PY
import json
l=[x for x in open("$R/$label.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$label", "factorize %.3f"%d["ms_per_factorize"], "solve %.3f"%d["ms_per_solve"], "it/s %.2f"%d["value"])
else: print("$label FAILED", open("$R/$label.log").read()[-300:])
PY
}
run base X=1 --
run base2 X=1 --
run nb8tail3k MNK_PP_NB8_ROWS=3072 --
run nb8tail5k MNK_PP_NB8_ROWS=5120 --
run nb8tail7k MNK_PP_NB8_ROWS=7168 --
run own128 MNK_OWN_COLS=128 --
run own192 MNK_OWN_COLS=192 --
run own256 MNK_OWN_COLS=256 --
run own128_nb8tail5k MNK_OWN_COLS=128 MNK_PP_NB8_ROWS=5120 --
run mid300 MNK_SMALL_TILES_MID=300 --
run mid300_own128 MNK_SMALL_TILES_MID=300 MNK_OWN_COLS=128 --

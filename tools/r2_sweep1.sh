#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2d
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
run nbo1024 X=1 -- --outer-block 1024
run nbo768 X=1 -- --outer-block 768
run nbo256 X=1 -- --outer-block 256
run tail256_5k MNK_TAIL_ROWS=5120 MNK_TAIL_NBO=256 --
run tail256_3k MNK_TAIL_ROWS=3072 MNK_TAIL_NBO=256 --
run mid0 MNK_SMALL_TILES_MID=0 --
run mid100 MNK_SMALL_TILES_MID=100 --
run mid300 MNK_SMALL_TILES_MID=300 --
run nbo1024_tail MNK_TAIL_ROWS=5120 MNK_TAIL_NBO=512 -- --outer-block 1024
run nbo1024_mid100 MNK_SMALL_TILES_MID=100 -- --outer-block 1024
run chol X=1 -- --algorithm CHOLESKY

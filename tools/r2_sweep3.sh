#!/bin/bash
# knob sweep with the persistent panel kernel as the default panel step (round 2, second half)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2e
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
run algo1 MNK_PANEL_ALGO=1 --
run nb8 MNK_PP_NB=8 --
run nbo1024 X=1 -- --outer-block 1024
run nbo768 X=1 -- --outer-block 768
run nbo256 X=1 -- --outer-block 256
run nbo256_sa0 MNK_SPLIT_A=0 -- --outer-block 256
run sa0 MNK_SPLIT_A=0 --
run sa1 MNK_SPLIT_A=1 --
run tail256_5k MNK_TAIL_ROWS=5120 MNK_TAIL_NBO=256 --
run tail256_3k MNK_TAIL_ROWS=3072 MNK_TAIL_NBO=256 --
run tail256_5k_sa0 MNK_TAIL_ROWS=5120 MNK_TAIL_NBO=256 MNK_SPLIT_A=0 --
run pcus32 MNK_PANEL_CUS=32 --
run pcus48 MNK_PANEL_CUS=48 --
run pcus80 MNK_PANEL_CUS=80 --
run pcus96 MNK_PANEL_CUS=96 --
run mid0 MNK_SMALL_TILES_MID=0 --
run mid300 MNK_SMALL_TILES_MID=300 --
run p0part MNK_PANEL0_WHOLE=0 --
run share2 MNK_SHARE=2 --
run chol X=1 -- --algorithm CHOLESKY

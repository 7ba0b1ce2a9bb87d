#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2h
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
run() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
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
run st0 MNK_SMALL_TILES=0 --
run st100 MNK_SMALL_TILES=100 --
run st200 MNK_SMALL_TILES=200 --
run st300 MNK_SMALL_TILES=300 --
run split1 MNK_SPLIT_A=1 --
run split1_st0 MNK_SPLIT_A=1 MNK_SMALL_TILES=0 --
run pcu48 MNK_PANEL_CUS=48 --
run pcu80 MNK_PANEL_CUS=80 --
run pcu96 MNK_PANEL_CUS=96 --
run share2 MNK_SHARE=2 --
run base2 X=1 --

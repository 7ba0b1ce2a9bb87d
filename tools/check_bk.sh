#!/bin/bash
# GPU check of the pivoted (Bunch-Kaufman) tier: its tests, then timings of one factorization at N = 4000 / 11192 with the
# multi-workgroup panels and with one workgroup per panel.  -> gpurun_out/chk_bk
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/chk_bk
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_round4.py -x -q -m gpu -k "bunchkaufman" -s > $R/t_new.log 2>&1; tail -6 $R/t_new.log
timeout 600 python -m pytest tests/test_hip_round2.py tests/test_hip_parity.py tests/test_hip_stress.py -x -q -m gpu -k "bunchkaufman or growth or pivot or tier or dense_kkt or bk" -s > $R/t_old.log 2>&1; tail -4 $R/t_old.log
for n in 4000 11192; do
  timeout 120 python tools/bk_run.py $n 0 2>&1 | tail -1
  timeout 120 python tools/bk_run.py $n 1 2>&1 | tail -1
done

#!/bin/bash
# Kernel trace of one factorization that takes the pivoted tier (tools/bk_run.py): per-kernel totals -> gpurun_out/bk_trace
export TMPDIR=/tmp
N=${1:-11192}
R=$GRAFT_REPO_ROOT/gpurun_out/bk_trace
rm -rf $R; mkdir -p $R
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/t -o p -- python $GRAFT_REPO_ROOT/tools/bk_run.py $N 0 > $R/run.log 2>&1
tail -2 $R/run.log
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $R/t -name "*.db" | head -1) $R/bk_kernel_stats_N$N.md | head -30
rm -rf $R/t

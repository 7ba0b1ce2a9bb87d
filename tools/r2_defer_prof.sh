#!/bin/bash
# kernel statistics of the bench command with a given MNK_DEFER_ROWS (arg 1) -> gpurun_out/defer_stats_<rows>.md
export TMPDIR=/tmp
d=${1:-5120}
R=$GRAFT_REPO_ROOT/gpurun_out/prof_defer
rm -rf $R; mkdir -p $R
cd /tmp
MNK_DEFER_ROWS=$d timeout 300 rocprofv3 --kernel-trace -d $R/bench -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-ipm-loop --steps 10 --warmup 2 > $R/log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $R/bench -name "*.db" | head -1) gpurun_out/defer_stats_$d.md > /dev/null
head -30 gpurun_out/defer_stats_$d.md
rm -rf $R

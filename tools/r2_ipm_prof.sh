#!/bin/bash
# kernel statistics of the device-resident IPM loop (tools/ipm_run_device.py) -> gpurun_out/r02_ipm_loop_kernel_stats.md
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_ipm
rm -rf $R; mkdir -p $R
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/t -o p -- python $GRAFT_REPO_ROOT/tools/ipm_run_device.py ${1:-case1354pegase} > $R/log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $R/t -name "*.db" | head -1) gpurun_out/r02_ipm_loop_kernel_stats.md > /dev/null
head -45 gpurun_out/r02_ipm_loop_kernel_stats.md | cut -c1-160
grep '^{' $R/log | tail -2 | cut -c1-400
rm -rf $R

#!/bin/bash
# kernel statistics of the device-resident IPM loop alone (tools/ipm_run_device.py, IPM_DEVICE_ONLY) on the AC-OPF NLP
# -> gpurun_out/r03_ipm_loop_kernel_stats.md        usage: tools/r3_ipm_prof.sh [grid]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_ipm
rm -rf $R; mkdir -p $R
cd /tmp
IPM_DEVICE_ONLY=1 timeout 400 rocprofv3 --kernel-trace -d $R/t -o p -- python $GRAFT_REPO_ROOT/tools/ipm_run_device.py acopf ${1:-case1354pegase} > $R/log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $R/t -name "*.db" | head -1) gpurun_out/r03_ipm_loop_kernel_stats.md > /dev/null
head -60 gpurun_out/r03_ipm_loop_kernel_stats.md | cut -c1-160
grep -c "at::native\|rocblas" gpurun_out/r03_ipm_loop_kernel_stats.md
grep '^{' $R/log | tail -2 | cut -c1-500
grep '^{' $R/log > gpurun_out/r03_ipm_run_device_resident_acopf_case1354.jsonl
tail -3 $R/log | cut -c1-300
rm -rf $R/t

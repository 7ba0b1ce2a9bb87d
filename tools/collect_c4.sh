#!/bin/bash
# Run on the GPU box: kernel trace + MFMA-busy PMC pass of the C4-size (N = 85568) factorization -> gpurun_out/prof_c4
export TMPDIR=/tmp
R=/root/repo/gpurun_out/prof_c4
rm -rf $R; mkdir -p $R
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $R/trace -o p -- python /root/repo/tools/bench_configs.py c4 > $R/trace.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/mfma -o p -- python /root/repo/tools/bench_configs.py c4 > $R/mfma.log 2>&1
grep config $R/trace.log | cut -c1-300
grep config $R/mfma.log | cut -c1-300
ls -la $R/*/

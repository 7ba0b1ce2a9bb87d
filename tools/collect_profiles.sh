#!/bin/bash
# Run on the GPU box: kernel trace of bench.py + PMC passes of one C3-size factorization -> gpurun_out/
set -x
export TMPDIR=/tmp
R=/root/repo/gpurun_out/prof_final
rm -rf $R; mkdir -p $R
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/bench -o p -- python /root/repo/bench.py --steps 10 --warmup 2 > $R/bench.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/fetch -o p -- python /root/repo/tools/prof_factor.py 11192 LDL 512 1 > $R/fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/write -o p -- python /root/repo/tools/prof_factor.py 11192 LDL 512 1 > $R/write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/mfma -o p -- python /root/repo/tools/prof_factor.py 11192 LDL 512 1 > $R/mfma.log 2>&1
ls -la $R/*/
tail -2 $R/bench.log | cut -c1-300

#!/bin/bash
# Round-6 evidence for BASELINE config C4 (N = 85 568, schedule 4: one launch per panel piece -- it serializes cleanly under --pmc), run on
# the GPU box: kernel trace, MFMA-busy pass, FETCH_SIZE / WRITE_SIZE passes (separate, counters + kernel trace only) of
# tools/bench_configs.py c4 -> gpurun_out/prof_c4_r06/{r06_config_C4_kernel_stats.md, r06_config_C4_pmc.md, r06_config_C4_pmc_traffic.{md,json}}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_c4_r06
rm -rf $R; mkdir -p $R
cd /tmp
C="python $GRAFT_REPO_ROOT/tools/bench_configs.py c4"
timeout 500 rocprofv3 --kernel-trace -d $R/trace -o p -- $C > $R/trace.log 2>&1
timeout 700 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/mfma -o p -- $C > $R/mfma.log 2>&1
timeout 700 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/fetch -o p -- $C > $R/fetch.log 2>&1
timeout 700 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/write -o p -- $C > $R/write.log 2>&1
db() { find $R/$1 -name "*.db" | head -1; }
cd $GRAFT_REPO_ROOT
for f in trace mfma fetch write; do grep '"config"' $R/$f.log | cut -c1-400; done
python tools/rocpd_stats.py $(db trace) $R/r06_config_C4_kernel_stats.md | tail -3
CFG="tools/bench_configs.py c4: case9241pegase-shaped sparse condensed KKT, N=85568, BUNCHKAUFMAN tier 1 (LDL^T), schedule 4 (one launch per panel piece), round-6 build"
python tools/pmc_report.py mfma $(db mfma) $R/r06_config_C4_pmc.md "$CFG" | tail -15
python tools/pmc_report.py traffic $(db fetch) $(db write) $R/r06_config_C4_pmc_traffic.md $R/r06_config_C4_pmc_traffic.json "$CFG" | tail -8
rm -rf $R/trace $R/mfma $R/fetch $R/write
ls -la $R

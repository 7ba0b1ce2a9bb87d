#!/bin/bash
# PMC passes of bench.py with the launch-per-panel schedule (panel_algo 4) -> gpurun_out/prof_r03 (see collect_profiles_r03.sh)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r03
mkdir -p $R
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-ipm-loop --no-c4"
MNK_PANEL_ALGO=4 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/fetch -o p -- $B --steps 2 --warmup 1 > $R/fetch.log 2>&1
MNK_PANEL_ALGO=4 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/write -o p -- $B --steps 2 --warmup 1 > $R/write.log 2>&1
MNK_PANEL_ALGO=4 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/mfma -o p -- $B --steps 2 --warmup 1 > $R/mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $R/bench -o p -- $B --steps 10 --warmup 2 > $R/bench_under_rocprof.log 2>&1
db() { find $R/$1 -name "*.db" | head -1; }
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(db bench) $R/r03_bench_kernel_stats.md | tail -3
python tools/pmc_report.py traffic $(db fetch) $(db write) $R/r03_pmc_traffic_factorize_N11192_panel_algo4.md $R/r03_pmc_traffic_panel_algo4.json "bench.py C3 with MNK_PANEL_ALGO=4 (one launch per 256-column panel + one trailing update per outer panel): case1354pegase-shaped sparse condensed KKT, N=11192, BUNCHKAUFMAN tier 1 (LDL^T)" | tail -3
python tools/pmc_report.py mfma $(db mfma) $R/r03_pmc_mfma_panel_algo4.md "bench.py C3 with MNK_PANEL_ALGO=4, N=11192" | tail -12
grep '^{' $R/mfma.log | tail -1 | cut -c1-300
rm -rf $R/bench $R/fetch $R/write $R/mfma

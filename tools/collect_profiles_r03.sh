#!/bin/bash
# Round-3 records, run on the GPU box (-> gpurun_out/prof_r03, copied into profiles/ by hand):
#   kernel trace of bench.py (same command as the bench line), bench lines (C3, C5 batch, C4 inside the C3 line),
#   C2, the launch count of the blocked Bunch-Kaufman tier at N = 4000, the DAG schedule's own timeline / slot accounting,
#   PMC passes with the launch-per-panel schedule (panel_algo 4): rocprofv3's counter mode serializes kernels, the two
#   persistent kernels of the task-DAG schedule cannot run under it (they wait for each other).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r03
rm -rf $R; mkdir -p $R
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-ipm-loop --no-c4"
timeout 300 rocprofv3 --kernel-trace -d $R/bench -o p -- $B --steps 10 --warmup 2 > $R/bench_under_rocprof.log 2>&1
if [ -z "$SKIP_PMC" ]; then
MNK_PANEL_ALGO=4 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/fetch -o p -- $B --steps 2 --warmup 1 > $R/fetch.log 2>&1
MNK_PANEL_ALGO=4 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/write -o p -- $B --steps 2 --warmup 1 > $R/write.log 2>&1
MNK_PANEL_ALGO=4 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/mfma -o p -- $B --steps 2 --warmup 1 > $R/mfma.log 2>&1
fi
timeout 200 rocprofv3 --kernel-trace -d $R/bk -o p -- python $GRAFT_REPO_ROOT/tools/bk_run.py 4000 > $R/bk.log 2>&1
db() { find $R/$1 -name "*.db" | head -1; }
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(db bench) $R/r03_bench_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $(db bk) $R/r03_bunchkaufman_N4000_kernel_stats.md > /dev/null
if [ -z "$SKIP_PMC" ]; then
python tools/pmc_report.py traffic $(db fetch) $(db write) $R/r03_pmc_traffic_factorize_N11192_panel_algo4.md $R/r03_pmc_traffic_panel_algo4.json "bench.py C3 with MNK_PANEL_ALGO=4 (one launch per 256-column panel + one trailing update per outer panel): case1354pegase-shaped sparse condensed KKT, N=11192, BUNCHKAUFMAN tier 1 (LDL^T)" | tail -3
python tools/pmc_report.py mfma $(db mfma) $R/r03_pmc_mfma_panel_algo4.md "bench.py C3 with MNK_PANEL_ALGO=4, N=11192" | tail -12
fi
grep '^{' $R/bench_under_rocprof.log | tail -1 > $R/r03_bench_N1_under_rocprof.json
python bench.py --steps 20 --warmup 5 > $R/r03_bench_N1.log 2>&1; grep '^{' $R/r03_bench_N1.log | tail -1 > $R/r03_bench_N1.json; cut -c1-400 $R/r03_bench_N1.json
python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > $R/r03_config_C5_batch16_per_gpu.json; cut -c1-300 $R/r03_config_C5_batch16_per_gpu.json
python tools/bench_configs.py c2 2>&1 | grep '^{' > $R/r03_config_C2_dense_condensed.jsonl; cat $R/r03_config_C2_dense_condensed.jsonl | cut -c1-400
python tools/ipm_run_device.py dense 2048 512 0 2>&1 | grep '^{' > $R/r03_ipm_run_device_resident_C2_dense.jsonl
python tools/dag_chain.py 11192 LDL > $R/dag_chain_C3.txt 2>&1
python tools/dag_util.py 11192 LDL > $R/dag_util_C3.txt 2>&1
rm -rf $R/bench $R/fetch $R/write $R/mfma $R/bk
tail -5 $R/r03_bench_kernel_stats.md; grep -c . $R/r03_bunchkaufman_N4000_kernel_stats.md; tail -3 $R/bk.log

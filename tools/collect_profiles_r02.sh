#!/bin/bash
# Run on the GPU box: kernel trace of bench.py + PMC passes of the SAME command (the bench's _sc densify path),
# counters in their own passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section) -> gpurun_out/prof_r02
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r02
[ -z "$SKIP_BENCH" ] && rm -rf $R; mkdir -p $R
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-ipm-loop"
timeout 300 rocprofv3 --kernel-trace -d $R/bench -o p -- $B --steps 10 --warmup 2 > $R/bench_under_rocprof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/fetch -o p -- $B --steps 2 --warmup 1 > $R/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/write -o p -- $B --steps 2 --warmup 1 > $R/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/mfma -o p -- $B --steps 2 --warmup 1 > $R/mfma.log 2>&1
db() { find $R/$1 -name "*.db" | head -1; }
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(db bench) $R/r02_bench_kernel_stats.md > /dev/null
python tools/pmc_report.py traffic $(db fetch) $(db write) $R/r02_pmc_traffic_factorize_N11192.md $R/r02_pmc_traffic.json "bench.py C3: case1354pegase-shaped sparse condensed KKT, N=11192, BUNCHKAUFMAN tier 1 (LDL^T), NBO=512, CSC->dense densify path" | tail -3
python tools/pmc_report.py mfma $(db mfma) $R/r02_pmc_mfma_utilization.md "bench.py C3, N=11192" | tail -12
grep '^{' $R/bench_under_rocprof.log | tail -1 > $R/r02_bench_N1_under_rocprof.json
if [ -z "$SKIP_BENCH" ]; then
python bench.py --steps 20 --warmup 5 > $R/r02_bench_N1.log 2>&1; grep '^{' $R/r02_bench_N1.log | tail -1 > $R/r02_bench_N1.json; cut -c1-600 $R/r02_bench_N1.json
python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > $R/r02_config_C5_batch16_per_gpu.json; cut -c1-400 $R/r02_config_C5_batch16_per_gpu.json
fi
rm -rf $R/bench $R/fetch $R/write $R/mfma
tail -5 $R/r02_bench_kernel_stats.md

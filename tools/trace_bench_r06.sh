export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r06b
rm -rf $R; mkdir -p $R
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-ipm-loop --no-c4 --no-c5-shape --no-live-traffic"
timeout 300 rocprofv3 --kernel-trace -d $R/bench -o p -- $B --steps 10 --warmup 2 > $R/bench_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $R/bench -name "*.db" | head -1) $R/r06_bench_kernel_stats.md | tail -3
grep '^{' $R/bench_under_rocprof.log | tail -1 > $R/r06_bench_N1_under_rocprof.json
rm -rf $R/bench
head -8 $R/r06_bench_kernel_stats.md; tail -3 $R/r06_bench_kernel_stats.md; cut -c1-400 $R/r06_bench_N1_under_rocprof.json

#!/bin/bash
# Round-5 records, run on the GPU box (-> gpurun_out/prof_r05, copied into profiles/ by hand):
#   kernel trace of bench.py (same command as the bench line), bench lines (C3; C5 shape on one GPU with the batch API and
#   over 4 contexts), C2, counters of the task-DAG schedule through the device counting service (regenerated:
#   roofline.traffic of the bench line points at r05_pmc_traffic.json), the DAG's own timeline, the IPM loop's kernel trace.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r05
rm -rf $R; mkdir -p $R
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-ipm-loop --no-c4"
timeout 300 rocprofv3 --kernel-trace -d $R/bench -o p -- $B --steps 10 --warmup 2 > $R/bench_under_rocprof.log 2>&1
db() { find $R/$1 -name "*.db" | head -1; }
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(db bench) $R/r05_bench_kernel_stats.md > /dev/null
grep '^{' $R/bench_under_rocprof.log | tail -1 > $R/r05_bench_N1_under_rocprof.json
rm -rf $R/bench
export ROCP_TOOL_LIBRARIES=$GRAFT_REPO_ROOT/tools/devcount/libmnk_devcount.so
for s in mfma fetch write; do timeout 200 python tools/devcount_dag.py $s 20 > $R/dc_$s.json 2> $R/dc_$s.err; done
unset ROCP_TOOL_LIBRARIES
python tools/devcount_report.py $R/dc_mfma.json $R/dc_fetch.json $R/dc_write.json $R/r05_pmc_dag_C3.md $R/r05_pmc_traffic.json | tail -12
cp $R/r05_pmc_traffic.json profiles/r05_pmc_traffic.json 2>/dev/null   # (the bench line below reads it)
python bench.py --steps 20 --warmup 5 > $R/r05_bench_N1.log 2>&1; grep '^{' $R/r05_bench_N1.log | tail -1 > $R/r05_bench_N1.json; cut -c1-300 $R/r05_bench_N1.json
python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline --no-c4 --no-ipm-loop 2>&1 | grep '^{' | tail -1 > $R/r05_config_C5_batch16_per_gpu.json; cut -c1-200 $R/r05_config_C5_batch16_per_gpu.json
python bench.py --steps 5 --warmup 2 --batch 16 --concurrency 4 --no-cpu-baseline --no-c4 --no-ipm-loop 2>&1 | grep '^{' | tail -1 > $R/r05_config_C5_batch16_4contexts.json; cut -c1-200 $R/r05_config_C5_batch16_4contexts.json
python tools/bench_configs.py c2 2>&1 | grep '^{' > $R/r05_config_C2_dense_condensed.jsonl; cut -c1-300 $R/r05_config_C2_dense_condensed.jsonl
python tools/dag_chain.py 11192 LDL > $R/r05_dag_chain_C3.txt 2>&1
python tools/dag_util.py 11192 LDL > $R/r05_dag_util_C3.txt 2>&1
python tools/bench_small_batch.py > $R/r05_small_batches.txt 2>&1; tail -4 $R/r05_small_batches.txt | cut -c1-250
# the device-resident IPM loop under the kernel trace: back-solve / factorization time per interior-point iteration
cd /tmp
IPM_DEVICE_ONLY=1 timeout 300 rocprofv3 --kernel-trace -d $R/ipm -o p -- python $GRAFT_REPO_ROOT/tools/ipm_run_device.py acopf case1354pegase > $R/ipm_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(db ipm) $R/r05_ipm_loop_kernel_stats.md > /dev/null
grep '^{' $R/ipm_under_rocprof.log | tail -2 > $R/r05_ipm_run_device_resident_acopf_case1354.jsonl
rm -rf $R/ipm
ls $R

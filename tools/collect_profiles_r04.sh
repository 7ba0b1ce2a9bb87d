#!/bin/bash
# Round-4 records, run on the GPU box (-> gpurun_out/prof_r04, copied into profiles/ by hand):
#   kernel trace of bench.py (same command as the bench line), bench lines (C3; C5 shape on one GPU: batch API, without it,
#   4 contexts), C2, counters of the task-DAG schedule through the device counting service, the DAG's own timeline.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r04
rm -rf $R; mkdir -p $R
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-ipm-loop --no-c4"
timeout 300 rocprofv3 --kernel-trace -d $R/bench -o p -- $B --steps 10 --warmup 2 > $R/bench_under_rocprof.log 2>&1
db() { find $R/$1 -name "*.db" | head -1; }
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(db bench) $R/r04_bench_kernel_stats.md > /dev/null
grep '^{' $R/bench_under_rocprof.log | tail -1 > $R/r04_bench_N1_under_rocprof.json
rm -rf $R/bench
python bench.py --steps 20 --warmup 5 > $R/r04_bench_N1.log 2>&1; grep '^{' $R/r04_bench_N1.log | tail -1 > $R/r04_bench_N1.json; cut -c1-300 $R/r04_bench_N1.json
python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > $R/r04_config_C5_batch16_per_gpu.json; cut -c1-200 $R/r04_config_C5_batch16_per_gpu.json
python bench.py --steps 5 --warmup 2 --batch 16 --concurrency 4 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > $R/r04_config_C5_batch16_4contexts.json; cut -c1-200 $R/r04_config_C5_batch16_4contexts.json
python bench.py --steps 5 --warmup 2 --batch 16 --no-batch-api --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > $R/r04_config_C5_batch16_no_batch_api.json; cut -c1-200 $R/r04_config_C5_batch16_no_batch_api.json
python tools/bench_configs.py c2 2>&1 | grep '^{' > $R/r04_config_C2_dense_condensed.jsonl; cut -c1-300 $R/r04_config_C2_dense_condensed.jsonl
bash tools/batch_trace.sh > $R/r04_batch_step_timeline.txt 2>&1; tail -8 $R/r04_batch_step_timeline.txt
python tools/dag_chain.py 11192 LDL > $R/dag_chain_C3.txt 2>&1
python tools/dag_util.py 11192 LDL > $R/dag_util_C3.txt 2>&1
export ROCP_TOOL_LIBRARIES=$GRAFT_REPO_ROOT/tools/devcount/libmnk_devcount.so
for s in mfma fetch write; do timeout 200 python tools/devcount_dag.py $s 20 > $R/dc_$s.json 2> $R/dc_$s.err; done
unset ROCP_TOOL_LIBRARIES
python tools/devcount_report.py $R/dc_mfma.json $R/dc_fetch.json $R/dc_write.json $R/r04_pmc_dag_C3.md $R/r04_pmc_traffic.json | tail -12
# small-system batches, the Schur stage, the pivoted tier
python tools/bench_small_batch.py > $R/small_batches.txt 2>&1; tail -4 $R/small_batches.txt | cut -c1-250
python tools/bench_schur.py 2>/dev/null | grep '^{' > $R/r04_schur_stage.jsonl; python tools/bench_schur.py 128 512 256 2>/dev/null | grep '^{' >> $R/r04_schur_stage.jsonl; cut -c1-200 $R/r04_schur_stage.jsonl
for n in 4000 11192; do for w in 0 1; do timeout 120 python tools/bk_run.py $n $w 2>/dev/null | tail -1; done; done > $R/bunchkaufman_times.txt; cat $R/bunchkaufman_times.txt

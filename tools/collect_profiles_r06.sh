#!/bin/bash
# Round-6 records, run on the GPU box (-> gpurun_out/prof_r06, copied into profiles/ by hand):
#   kernel trace of bench.py (same command as the bench line), bench lines (C3; C5 shape on one GPU with the batch API and
#   over 4 contexts), C2, counters of the task-DAG schedule through the device counting service (regenerated:
#   roofline.traffic of the bench line points at r06_pmc_traffic.json), the DAG's own timeline, the IPM loop's kernel trace.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r06
rm -rf $R; mkdir -p $R
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-ipm-loop --no-c4 --no-c5-shape --no-live-traffic"
timeout 300 rocprofv3 --kernel-trace -d $R/bench -o p -- $B --steps 10 --warmup 2 > $R/bench_under_rocprof.log 2>&1
db() { find $R/$1 -name "*.db" | head -1; }
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(db bench) $R/r06_bench_kernel_stats.md > /dev/null
grep '^{' $R/bench_under_rocprof.log | tail -1 > $R/r06_bench_N1_under_rocprof.json
rm -rf $R/bench
export ROCP_TOOL_LIBRARIES=$GRAFT_REPO_ROOT/tools/devcount/libmnk_devcount.so
for s in mfma fetch write; do timeout 200 python tools/devcount_dag.py $s 20 > $R/dc_$s.json 2> $R/dc_$s.err; done
unset ROCP_TOOL_LIBRARIES
python tools/devcount_report.py $R/dc_mfma.json $R/dc_fetch.json $R/dc_write.json $R/r06_pmc_dag_C3.md $R/r06_pmc_traffic.json | tail -12
cp $R/r06_pmc_traffic.json profiles/r06_pmc_traffic.json 2>/dev/null   # (the bench line below reads it)
python bench.py --steps 20 --warmup 5 > $R/r06_bench_N1.log 2>&1; grep '^{' $R/r06_bench_N1.log | tail -1 > $R/r06_bench_N1.json; cut -c1-300 $R/r06_bench_N1.json
python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline --no-c4 --no-ipm-loop 2>&1 | grep '^{' | tail -1 > $R/r06_config_C5_batch16_per_gpu.json; cut -c1-200 $R/r06_config_C5_batch16_per_gpu.json
python bench.py --steps 5 --warmup 2 --batch 16 --concurrency 4 --no-cpu-baseline --no-c4 --no-ipm-loop 2>&1 | grep '^{' | tail -1 > $R/r06_config_C5_batch16_4contexts.json; cut -c1-200 $R/r06_config_C5_batch16_4contexts.json
python tools/bench_configs.py c2 2>&1 | grep '^{' > $R/r06_config_C2_dense_condensed.jsonl; cut -c1-300 $R/r06_config_C2_dense_condensed.jsonl
python tools/dag_chain.py 11192 LDL > $R/r06_dag_chain_C3.txt 2>&1
python tools/dag_util.py 11192 LDL > $R/r06_dag_util_C3.txt 2>&1
python tools/bench_small_batch.py > $R/r06_small_batches.txt 2>&1; tail -4 $R/r06_small_batches.txt | cut -c1-250
# the pivot leaf on its own, the instruction kinds it is made of, the chain strips' tile step, the chain's hand-over and per-step time line
./tools/hip/leaf_lab > $R/r06_leaf_lab.txt 2>&1
./tools/hip/valu_lat > $R/r06_valu_latencies.txt 2>&1
./tools/hip/mfma_dep >> $R/r06_valu_latencies.txt 2>&1
./tools/hip/tilestep_lab > $R/r06_tilestep_lab.txt 2>&1
(python tools/chain_steps.py 2048 LDL; MNK_LIBPATH=madnlp.jl_amd/lib/libmadnlp_hip_leaf0.so python tools/chain_steps.py 2048 LDL) 2>&1 | grep -v amdgpu.ids | grep "mean over\|factorize" > $R/r06_chain_handover.txt
MNK_LIBPATH=madnlp.jl_amd/lib/libmadnlp_hip_steptr.so python tools/chain_steps2.py 2048 LDL 3 2>&1 | grep -v amdgpu.ids > $R/r06_chain_steps_dma.txt
bash tools/leaf_ab.sh leaf0 2048 6100 11192 > /dev/null 2>&1; cp gpurun_out/leaf_ab_leaf0.txt $R/r06_leaf_ab.txt
python tools/spec_pair_time.py 2>&1 | grep -v amdgpu.ids > $R/r06_spec_pair_time.txt
# the device-resident IPM loop under the kernel trace: back-solve / factorization time per interior-point iteration
cd /tmp
IPM_DEVICE_ONLY=1 timeout 300 rocprofv3 --kernel-trace -d $R/ipm -o p -- python $GRAFT_REPO_ROOT/tools/ipm_run_device.py acopf case1354pegase > $R/ipm_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(db ipm) $R/r06_ipm_loop_kernel_stats.md > /dev/null
grep '^{' $R/ipm_under_rocprof.log | tail -2 > $R/r06_ipm_run_device_resident_acopf_case1354.jsonl
rm -rf $R/ipm
ls $R

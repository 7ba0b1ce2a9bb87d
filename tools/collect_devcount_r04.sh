#!/bin/bash
# Counter passes of the task-DAG schedule (device counting service) -> gpurun_out/devcount_r04 (copied into profiles/ by hand)
cd $GRAFT_REPO_ROOT
O=gpurun_out/devcount_r04; rm -rf $O; mkdir -p $O
export ROCP_TOOL_LIBRARIES=$GRAFT_REPO_ROOT/tools/devcount/libmnk_devcount.so
for s in mfma fetch write; do timeout 200 python tools/devcount_dag.py $s 20 > $O/dc_$s.json 2> $O/dc_$s.err; done
unset ROCP_TOOL_LIBRARIES
python tools/devcount_report.py $O/dc_mfma.json $O/dc_fetch.json $O/dc_write.json $O/r04_pmc_dag_C3.md $O/r04_pmc_traffic.json | tail -20

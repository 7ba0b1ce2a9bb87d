#!/bin/bash
# Quick GPU check of a change to the pivot-chain kernels: the tests that exercise the chain (range / determinism, batches,
# small batches, stress), the chain's own timeline, and the bench line.  -> gpurun_out/chk
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/chk
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_round4.py tests/test_hip_c5.py tests/test_hip_stress.py tests/test_schur.py -x -q -m gpu > $R/tests.log 2>&1
tail -5 $R/tests.log
timeout 300 python -m pytest tests/test_hip_round2.py tests/test_hip_parity.py -x -q -m gpu > $R/tests2.log 2>&1
tail -3 $R/tests2.log
timeout 120 python tools/dag_chain.py 11192 LDL > $R/dag_chain_C3.txt 2>&1; head -3 $R/dag_chain_C3.txt; tail -6 $R/dag_chain_C3.txt | cut -c1-200
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ipm-loop --no-c4 2>&1 | grep '^{' | tail -1 > $R/bench.json; cut -c1-400 $R/bench.json
timeout 120 python tools/bench_configs.py c2 2>&1 | grep '^{' > $R/c2.jsonl; cut -c1-300 $R/c2.jsonl
timeout 120 python tools/bench_small_batch.py > $R/small.txt 2>&1; tail -8 $R/small.txt

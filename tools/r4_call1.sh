#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c1; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round4.py -x -q 2>&1 | tail -25 > $O/t_round4.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/t_full.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c4 2> $O/bench1.err | grep '^{' > $O/bench1.json
timeout 300 python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline 2> $O/c5_1.err | grep '^{' > $O/c5_conc1.json
timeout 300 python bench.py --steps 5 --warmup 2 --batch 16 --concurrency 4 --no-cpu-baseline 2> $O/c5_4.err | grep '^{' > $O/c5_conc4.json
timeout 400 bash tools/step_gaps.sh > $O/stepgaps.log 2>&1
cp gpurun_out/stepgaps/step.txt $O/ 2>/dev/null
tail -5 $O/t_round4.log; tail -5 $O/t_full.log; cut -c1-600 $O/bench1.json; cut -c1-300 $O/c5_conc1.json; cut -c1-300 $O/c5_conc4.json

#!/bin/bash
# last sweep of round 2, host thread pools bounded (no cgroup throttling): small knobs around the defaults
cd $GRAFT_REPO_ROOT
b() { timeout 120 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-ipm-loop "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('factorize %.3f  it/s %.2f' % (d['ms_per_factorize'], d['value']))"; }
echo "base"; b; b; b
echo "tail256_4k"; MNK_TAIL_ROWS=4096 MNK_TAIL_NBO=256 b; MNK_TAIL_ROWS=4096 MNK_TAIL_NBO=256 b
echo "mid300"; MNK_SMALL_TILES_MID=300 b; MNK_SMALL_TILES_MID=300 b
echo "fuse3072"; MNK_PP_FUSE_ROWS=3072 b
echo "fuse5120"; MNK_PP_FUSE_ROWS=5120 b
echo "own192"; MNK_OWN_COLS=192 b
echo "spin wait"; MNK_SPIN_WAIT=1 b
echo "chol"; b --algorithm CHOLESKY

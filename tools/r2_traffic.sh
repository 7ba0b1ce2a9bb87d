#!/bin/bash
# A/B of the lower-tile enumeration (MNK_SUPER_W = 1: column-by-column, 8: super-columns): time and FETCH/WRITE traffic
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r2t
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "gemm_nt or cholesky_vs_lapack or schedules or queue or dense_condensed_build" 2>&1 | tail -4
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline"
for sw in 8 1 4 16; do
  MNK_SUPER_W=$sw timeout 120 $B --steps 8 --warmup 2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sw=$sw factorize %.3f ms  it/s %.2f'%(d['ms_per_factorize'], d['value']))"
done
cd /tmp
for sw in 8 1; do
  MNK_SUPER_W=$sw timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/fetch$sw -o p -- $B --steps 2 --warmup 1 > $R/fetch$sw.log 2>&1
  MNK_SUPER_W=$sw timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/write$sw -o p -- $B --steps 2 --warmup 1 > $R/write$sw.log 2>&1
  F=$(find $R/fetch$sw -name "*.db" | head -1); W=$(find $R/write$sw -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_report.py traffic $F $W $R/traffic_sw$sw.md $R/traffic_sw$sw.json "bench.py C3 N=11192, MNK_SUPER_W=$sw" | tail -14
  rm -rf $R/fetch$sw $R/write$sw
done

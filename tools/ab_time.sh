#!/bin/bash
# Same-box A/B of factorize!: the tree's library against the copy under _ab/ (a build of another commit), alternating.
# usage: bash tools/ab_time.sh [N] [LDL|CHOLESKY] [rounds]
N=${1:-11192}; ALG=${2:-LDL}; ROUNDS=${3:-4}
cd $GRAFT_REPO_ROOT
for i in $(seq $ROUNDS); do
  echo -n "new: "; python tools/dag_time.py $N $ALG 2>/dev/null | tail -1
  echo -n "old: "; python _ab/tools/dag_time.py $N $ALG 2>/dev/null | tail -1
done

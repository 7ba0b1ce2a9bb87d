#!/bin/bash
# A/B of the schedule knobs on the C3-size factorization (run on the GPU box)
cd /root/repo
for cfg in "MNK_SPLIT_A=0 MNK_SHARE=0 MNK_SMALL_TILES=0" "MNK_SPLIT_A=0" "MNK_SPLIT_A=1" "MNK_SPLIT_A=1 MNK_SMALL_TILES_MID=400" "MNK_SPLIT_A=1 MNK_SMALL_TILES=800" "MNK_SPLIT_A=1 MNK_SHARE=0"; do
  echo "== $cfg"
  env $cfg python tools/prof_factor.py ${1:-11192} ${2:-LDL} 512 5 2>&1 | tail -3
done

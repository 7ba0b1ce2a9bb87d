#!/bin/bash
# A/B of the schedule knobs of the factorization (run on the GPU box):  tools/sweep_knobs.sh [N] [LDL|CHOLESKY]
# Each line is one environment override of the library defaults (see INTEGRATION.md for the list).
cd /root/repo
for cfg in "MNK_SHARE=1" "MNK_SHARE=0" "MNK_SPLIT_A=1" "MNK_SMALL_TILES=0 MNK_SMALL_TILES_MID=0" "MNK_NBM=256" \
           "MNK_PANEL_CUS=32" "MNK_PANEL_CUS=96" "MNK_LOOKAHEAD=0" "MNK_SINGLE_ROWS=0" "MNK_PERSISTENT_SOLVE=0"; do
  echo "== $cfg"
  env $cfg python tools/prof_factor.py ${1:-11192} ${2:-LDL} 512 5 2>&1 | grep factorize | tail -2
done

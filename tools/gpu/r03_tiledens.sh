#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_tiledens}
mkdir -p $O
cd $R
for d in 16 12 20 24 32 16; do
  echo "== NS_TILE_DENS64=$d" | tee -a $O/tiledens.txt
  NS_TILE_DENS64=$d timeout -k 10 300 python3 tools/law_bench.py --laws cfg5,cfg3,cfg5_2hot,cfg5_q2048 --reps 8 2>&1 | grep -v "^variant\|^  *law" | tee -a $O/tiledens.txt || exit 1
done
for d in 16 24; do
  echo "== NS_TILE_DENS64=$d big20" | tee -a $O/tiledens.txt
  NS_TILE_DENS64=$d timeout -k 10 400 python3 tools/law_bench.py --segments 20 --qscale 0.125 --laws cfg5 --reps 5 2>&1 | grep -v "^variant\|^  *law" | tee -a $O/tiledens.txt || exit 1
done

#!/bin/bash
# tile / general boundary again, now that the shared term scores made the tile body cheaper than the general one got
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_tiledens2}
mkdir -p $O
cd $R
for d in 16 12 10 8 14 16 12; do
  echo "== NS_TILE_DENS64=$d" | tee -a $O/tiledens.txt
  NS_TILE_DENS64=$d timeout -k 10 300 python3 tools/law_bench.py --laws cfg5,cfg5_seed7,cfg3,cfg5_q4096 --reps 8 2>&1 | grep -v "^variant\|^  *law" | cut -c1-100 | tee -a $O/tiledens.txt || exit 1
done

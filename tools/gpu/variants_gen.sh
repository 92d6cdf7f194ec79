#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-vgen}
mkdir -p $O
cd $R
for v in 0 12 13 14 15 16 17; do
  timeout -k 10 300 python3 tools/law_bench.py --variant $v --laws cfg5_gen,cfg5_2hot_gen,cfg5_1hot_gen,cfg5_thin >> $O/variants.txt 2>&1 || { tail -20 $O/variants.txt; exit 1; }
done
cat $O/variants.txt

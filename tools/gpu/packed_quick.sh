#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pq}
mkdir -p $O
cd $R
LAWS=cfg5,cfg5_thin,cfg5_gen,r1,r8
for opt in "" "--packed 1" "--impacts" "--packed 1 --impacts"; do
  timeout -k 10 300 python3 tools/law_bench.py $opt --laws $LAWS >> $O/law_small.txt 2>&1 || { tail -20 $O/law_small.txt; exit 1; }
done
for opt in "" "--packed 1"; do
  timeout -k 10 400 python3 tools/law_bench.py $opt --segments 20 --qscale 0.125 --laws $LAWS >> $O/law_big.txt 2>&1 || { tail -20 $O/law_big.txt; exit 1; }
done
cat $O/law_small.txt $O/law_big.txt

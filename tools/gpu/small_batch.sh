#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-sb}
mkdir -p $O
cd $R
for m in 0 1 2 3; do
  echo "NS_SMALL_SPLIT=$m" >> $O/split.txt
  NS_SMALL_SPLIT=$m timeout -k 10 300 python3 tools/law_bench.py --laws cfg5_q64,cfg5_q256,cfg5_q512,cfg5_q1024,cfg5_q2048,cfg5_q4096 >> $O/split.txt 2>&1 || { tail -20 $O/split.txt; exit 1; }
done
cat $O/split.txt

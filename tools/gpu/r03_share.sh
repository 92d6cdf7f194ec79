#!/bin/bash
# shared term scores: GPU suite, then the laws with sharing off / default / forced, alternating (same box)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_share}
mkdir -p $O
cd $R
if [ "${2:-tests}" = "tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
  tail -2 $O/tests.txt
fi
LAWS=${LAWS:-r1,cfg5_thin,cfg5_tile,cfg5_gen,cfg5,cfg5_seed7,cfg3,cfg3_k10,cfg5_q1024,cfg5_q2048,cfg5_q4096,hot5_k10}
for rep in 1 2; do
  for m in 0 1; do
    echo "== share $m (rep $rep)" >> $O/laws.txt
    timeout -k 10 300 python3 tools/law_bench.py --share $m --laws $LAWS --reps 8 2>&1 | grep -v "^variant\|^  *law" >> $O/laws.txt || exit 1
  done
done
cat $O/laws.txt

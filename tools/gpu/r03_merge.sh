#!/bin/bash
# full suite, then the merge body on / off (same library, NS_MERGE) on the laws it targets
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_merge}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider --durations=5 > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -4 $O/tests.txt
cd $R
for rep in 1 2; do
  for m in 0 1; do
    echo "== NS_MERGE=$m (rep $rep)" | tee -a $O/merge.txt
    NS_MERGE=$m timeout -k 10 300 python3 tools/law_bench.py --laws cfg5_t2_gen,r8r20,r8r300,cfg5_gen,cfg5,cfg3,cfg5_q2048 --reps 8 2>&1 | grep -v "^variant\|^  *law" | tee -a $O/merge.txt || exit 1
  done
done

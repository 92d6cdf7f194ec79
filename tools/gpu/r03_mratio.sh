#!/bin/bash
# two-list groups: up to which length ratio the merge body pays now that it reads the shared scores
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for r in 8 16 32 8 16 4; do
  echo "== NS_MERGE_RATIO=$r"
  NS_MERGE_RATIO=$r timeout -k 10 300 python3 tools/law_bench.py --laws cfg5,cfg5_seed7,cfg5_gen,r8r300,r8r20 --reps 8 2>&1 | grep -v "^variant\|^  *law" | cut -c1-90 || exit 1
done

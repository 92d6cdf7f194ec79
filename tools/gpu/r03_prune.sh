#!/bin/bash
# pruning parity tests + dealing-key experiment on the 1M-doc index (same box)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_prune}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider --durations=5 > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -9 $O/tests.txt
cd $R
for m in "0 3" "1 3" "2 3" "0 3" "1 3" "2 3"; do
  set -- $m
  echo "== mode $1 coarse $2 (1M docs)" | tee -a $O/deal1m.txt
  NS_ORDER_MODE=$1 NS_ORDER_COARSE=$2 timeout -k 10 300 python3 tools/law_bench.py --laws cfg5,cfg5_tile,cfg5_gen,cfg3,cfg5_q2048 --reps 8 2>&1 | grep -v "^variant\|^  *law" | tee -a $O/deal1m.txt || exit 1
done

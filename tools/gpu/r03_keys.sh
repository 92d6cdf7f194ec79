#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_keys}
mkdir -p $O
cd $R
for kp in "100,100,100,100" "90,100,100,100" "110,100,100,100" "100,90,100,100" "100,110,100,100" "100,100,90,100" "100,100,110,100" "100,100,100,80" "100,100,100,125" "85,100,100,100" "100,100,120,100" "100,100,100,100"; do
  echo "== NS_KEY_PCT=$kp" | tee -a $O/keys.txt
  NS_KEY_PCT=$kp timeout -k 10 300 python3 tools/law_bench.py --laws cfg5,cfg5_seed7,cfg5_q4096 --reps 8 2>&1 | grep -v "^variant\|^  *law" | tee -a $O/keys.txt || exit 1
done

#!/bin/bash
# long differential fuzz on the final library (three seeds) + the widening steps' fuzz + small-batch split sweep
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_fuzz}
mkdir -p $O
cd $R
for sp in 0 12288 16384 24576 32768 49152; do
  echo "== split $sp" | tee -a $O/split_q2048.txt
  timeout -k 10 200 python3 tools/law_bench.py --split $sp --laws cfg5_q2048,cfg5_q1024,cfg5_q4096 --reps 8 2>&1 | grep -v "^variant\|^  *law" | tee -a $O/split_q2048.txt || exit 1
done
for seed in 31 4711 271828; do
  timeout -k 10 400 python3 tools/fuzz_parity.py --seconds 180 --seed $seed 2>&1 | tail -2 | tee -a $O/fuzz_parity.txt || exit 1
done
timeout -k 10 300 python3 tools/fuzz_widening.py 150 2>&1 | tail -2 | tee -a $O/fuzz_widening.txt

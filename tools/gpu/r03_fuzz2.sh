#!/bin/bash
# more seeds of the differential fuzz on the final library
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_fuzz2}
mkdir -p $O
cd $R
for seed in ${SEEDS:-7 99 12345 777}; do
  timeout -k 10 400 python3 tools/fuzz_parity.py --seconds 200 --seed $seed 2>&1 | tail -1 | tee -a $O/fuzz_parity.txt || exit 1
done

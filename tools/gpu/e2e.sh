#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-e2e}
mkdir -p $O
cd $R
timeout -k 10 400 python3 tools/e2e_bench.py > $O/e2e.txt 2>&1 || { tail -20 $O/e2e.txt; exit 1; }
cat $O/e2e.txt

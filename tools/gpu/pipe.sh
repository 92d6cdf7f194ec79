#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pipe}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
cd $R
timeout -k 10 400 python3 tools/e2e_bench.py > $O/e2e.txt 2>&1 || { tail -20 $O/e2e.txt; exit 1; }
cat $O/e2e.txt
nproc; lscpu | grep "Model name"

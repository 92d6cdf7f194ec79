#!/bin/bash
# tests first (stop at the first failure), then the dealing / kernel A/B
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_full}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider --durations=5 > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -9 $O/tests.txt
cd $R
bash tools/gpu/r03_ab_deal.sh ${1:-r03_full}

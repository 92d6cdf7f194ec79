#!/bin/bash
# GPU box: the whole -m gpu suite on the working tree's libraries, then a same-box A/B of the HIP library against
# libnextsearch_hip_base.so (tools/dbg/build_ab.sh).   r03_tests_ab.sh <outdir-name> [laws] [pytest -k expr]
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_tests_ab}
LAWS=${2:-cfg5_thin,cfg5_tile,cfg5_gen,cfg5_2hot_gen,cfg5_1hot_gen,cfg5,cfg3,cfg5_q2048}
KEXPR=${3:-}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then
  timeout -k 10 1000 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider --durations=8 -k "$KEXPR" > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
else
  timeout -k 10 1000 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider --durations=8 > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
fi
tail -14 $O/tests.txt
cd $R
echo "#### A/B with the XCD dealing OFF in the new library (kernel change alone)"
NS_ORDER_MODE=0 bash tools/gpu/ab.sh ${1:-r03_tests_ab}_nodeal $LAWS
echo "#### A/B with the new library's defaults"
bash tools/gpu/ab.sh ${1:-r03_tests_ab} $LAWS

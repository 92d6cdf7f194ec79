#!/bin/bash
# GPU parity suite, then the skip-grid A/B (same box, flag on/off).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-skips}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
cd $R && bash tools/gpu/ab_flag.sh $1 ${2:-cfg5_tile,cfg5,cfg3,r1r2r3r4r5,cfg4} "--no-skips" ""

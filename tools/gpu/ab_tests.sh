#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-abt}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest $R/tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
cd $R && bash tools/gpu/ab.sh $1 ${2:-cfg3,cfg3_k64,r1r2r3r4r5,cfg5}

#!/bin/bash
# GPU box: the whole -m gpu suite, then the default bench line.  Usage: bash tools/gpu/tests_and_bench.sh <outdir-name>
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-tb}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider --durations=8 > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -14 $O/tests.txt
cd $R
timeout -k 10 400 python3 bench.py 2>$O/bench.err | tail -1 > $O/bench.json || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json

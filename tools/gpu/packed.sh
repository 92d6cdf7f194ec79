#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-packed}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest $R/tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "packed or digests or fuzz" > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
cd $R
LAWS=cfg5,cfg5_thin,cfg5_gen,cfg5_tile,r1,r8,r100,cfg3_k10
for opt in "" "--packed 1" "--packed 2" "--impacts" "--packed 1 --impacts"; do
  timeout -k 10 300 python3 tools/law_bench.py $opt --laws $LAWS >> $O/law_small.txt 2>&1 || { tail -20 $O/law_small.txt; exit 1; }
  timeout -k 10 400 python3 tools/law_bench.py $opt --segments 20 --qscale 0.125 --laws $LAWS >> $O/law_big.txt 2>&1 || { tail -20 $O/law_big.txt; exit 1; }
done
cat $O/law_small.txt $O/law_big.txt

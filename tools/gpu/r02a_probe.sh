#!/bin/bash
# Round-2 first probe: GPU tests at HEAD, then the query laws over a 20 x 1M-doc index (1.1 GB of postings +
# 0.55 GB of per-posting norms: beyond the 256 MiB Infinity Cache), with FETCH_SIZE of the same command.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider > $O/tests.txt 2>&1 || { tail -20 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
cd $R
LAWS=cfg5,cfg5_thin,cfg5_tile,cfg5_gen,scan_once,r1,r8,r100,r1000,cfg3_k10,cfg5_top1
timeout -k 10 600 python3 tools/law_bench.py --segments 20 --qscale 0.125 --laws $LAWS > $O/law_big.txt 2>&1 || { tail -20 $O/law_big.txt; exit 1; }
cat $O/law_big.txt
timeout -k 10 600 python3 tools/law_bench.py --segments 20 --qscale 0.125 --impacts --laws $LAWS > $O/law_big_imp.txt 2>&1 || { tail -20 $O/law_big_imp.txt; exit 1; }
cat $O/law_big_imp.txt
cd /tmp
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/law_bench.py --segments 20 --qscale 0.125 --laws cfg5,scan_once,cfg5_thin,cfg5_gen --reps 3 > $O/pmc_fetch.log 2>&1 || { tail -20 $O/pmc_fetch.log; exit 1; }
find $O/pmc_fetch -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/big_fetch_counter_collection.csv
rm -rf $O/pmc_fetch
echo done

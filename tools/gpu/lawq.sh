#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-lawq}
mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/law_bench.py --laws ${2:-r8,r16,r32,cfg5_thin,cfg5_tile,cfg5_gen,cfg5,cfg5_q64,cfg5_q256,cfg5_q512,cfg5_q1024,cfg5_q2048,cfg5_q4096,cfg3} > $O/law.txt 2>&1 || { tail -20 $O/law.txt; exit 1; }
cat $O/law.txt

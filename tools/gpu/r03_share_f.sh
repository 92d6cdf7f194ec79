#!/bin/bash
# pipelined value with sharing: what moves it (depth, overlap, steps)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_share_f}
mkdir -p $O
cd $R
one() {
  timeout -k 10 600 python3 bench.py --no-hbm-leg --no-impact-leg --cpu-seconds 0 "$@" > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  python3 -c "
import json,sys
d = json.load(open('$O/bench.json'))
print('$*'.ljust(44), 'value', round(d['value']), 'ms/step', round(d['ms_per_step'], 3), 'kernel_ms', round(d['roofline']['kernel_ms'], 3), 'kernel_only', round(d['kernel_only']['ms_per_step'], 3))"
}
for rep in 1 2; do
one --share 1
one --share 0
one --share 1 --steps 80
one --share 0 --steps 80
one --share 1 --depth 2
one --share 1 --depth 4
one --share 0 --depth 4
one --share 1 --no-overlap
one --share 0 --no-overlap
done

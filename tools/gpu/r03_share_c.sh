#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_share_c}
mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/dbg/prepare_time.py 16384,4096,2048 > $O/prepare.txt 2>&1 || { tail $O/prepare.txt; exit 1; }
cat $O/prepare.txt
for m in 1 0 1 0; do
  timeout -k 10 600 python3 bench.py --share $m --no-hbm-leg --no-impact-leg --cpu-seconds 0 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  python3 -c "
import json
d = json.load(open('$O/bench.json'))
print('share', $m, 'value', round(d['value']), 'ms/step', round(d['ms_per_step'], 3), 'kernel_ms', round(d['roofline']['kernel_ms'], 3), 'frac', round(d['roofline']['frac'], 4), 'kernel_only', round(d['kernel_only']['ms_per_step'], 3))"
done

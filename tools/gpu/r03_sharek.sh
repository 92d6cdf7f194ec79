#!/bin/bash
# duration of k_share_scores (rocprof kernel stats of the kernel leg), then the GPU suite
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_sharek}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/bench.py --kernel-only --steps 10 --warmup 2 --cpu-seconds 0 > $O/run.log 2>&1 || { tail $O/run.log; exit 1; }
grep -h "k_share_scores\|k_uscore" $(find $O/raw -name "*kernel_stats.csv" | head -1) | cut -c1-120
rm -rf $O/raw
cd $R && timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2

#!/bin/bash
# GPU box: event counts of the driver-stream body (diagnostic build swapped in on the box's scratch copy only)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-count}
mkdir -p $O
cd $R
# the product library is put back whatever happens (a later pytest / bench.py in this tree must never measure the counting build)
cp nextsearch-api_amd/libnextsearch_hip.so nextsearch-api_amd/libnextsearch_hip_product.so || exit 1
trap 'cp $R/nextsearch-api_amd/libnextsearch_hip_product.so $R/nextsearch-api_amd/libnextsearch_hip.so; rm -f $R/nextsearch-api_amd/libnextsearch_hip_product.so' EXIT
cp nextsearch-api_amd/libnextsearch_hip_count.so nextsearch-api_amd/libnextsearch_hip.so || exit 1
timeout -k 10 600 python3 tools/dbg/count_run.py ${2:-cfg5_gen,cfg5_2hot_gen,cfg5_1hot_gen,cfg5_nohot_gen,cfg5_thin,cfg5,cfg3} > $O/counts.txt 2>&1 || { tail -20 $O/counts.txt; exit 1; }
cat $O/counts.txt

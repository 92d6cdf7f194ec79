#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_sweep}
mkdir -p $O
cd $R
IDX=/tmp/ns_facade_idx
./nextsearch-api_amd/ns_tool gen-index $IDX 1 1000000 > /dev/null
python3 -c "
import sys; sys.path.insert(0,'nextsearch-api_amd'); import workloads
open('/tmp/cfg5.txt','w').write('\n'.join(workloads.cfg5_queries())+'\n')"
for sb in 2048 4096 5462 8192 100000; do
  echo "== NS_SUBBATCH=$sb" | tee -a $O/facade_sweep.txt
  NS_SUBBATCH=$sb timeout -k 10 120 ./nextsearch-api_amd/ns_tool facade-bench $IDX /tmp/cfg5.txt 10 11 0 | tee -a $O/facade_sweep.txt
done
echo "#### pruning on the single-term laws" | tee -a $O/prune.txt
for opt in "" "--prune"; do
  timeout -k 10 300 python3 tools/law_bench.py $opt --laws r1,r8,r32,r100,r1000,cfg5_top1,cfg5_thin,cfg5,cfg5_q2048 --reps 8 2>&1 | tee -a $O/prune.txt
done
for opt in "" "--prune"; do
  timeout -k 10 400 python3 tools/law_bench.py $opt --segments 20 --qscale 0.125 --laws r1,r8,cfg5_top1,cfg5_thin,cfg5 --reps 5 2>&1 | tee -a $O/prune.txt
done
echo "#### 20 x 1M docs, packed mode 1 (thin driver streams) with the dealing" | tee -a $O/packed.txt
for opt in "" "--packed 1"; do
  timeout -k 10 400 python3 tools/law_bench.py $opt --segments 20 --qscale 0.125 --laws cfg5,cfg5_thin,cfg5_gen,cfg5_tile --reps 5 2>&1 | tee -a $O/packed.txt
done

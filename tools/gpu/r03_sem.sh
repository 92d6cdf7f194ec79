#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_sem}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest $R/tests/test_semantic.py $R/tests/test_segment_shard.py -m gpu -x -q -p no:cacheprovider > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
cd $R
timeout -k 10 300 python3 tools/sem_bench.py > $O/sem_bench.json 2> $O/sem_bench.err || { tail $O/sem_bench.err; exit 1; }
cat $O/sem_bench.json
timeout -k 10 300 python3 tools/fuzz_widening.py 120 2>&1 | tail -2 | tee $O/fuzz_widening.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sem_stats -- python3 $R/tools/sem_bench.py > $O/sem_stats.log 2>&1 || { tail -5 $O/sem_stats.log; exit 1; }
cp $(find $O/sem_stats -name "*kernel_stats.csv" | head -1) $O/sem_kernel_stats.csv; rm -rf $O/sem_stats
head -6 $O/sem_kernel_stats.csv | cut -c1-80,200-330

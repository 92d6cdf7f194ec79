#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/$1 (run through gpurun from the repo root):
#   tests, bench lines of the four configs, kernel trace + stats of the cfg5 bench command, HBM traffic and
#   SQ counters of the same command (separate --pmc passes), the per-law table and the host-inclusive rates.
# tools/dbg/file_evidence.py <dir> then files the summaries under profiles/r01/ (run in the container).
set -o pipefail
D=${1:-ev}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$D
mkdir -p $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider > $O/tests.txt 2>&1 || { tail -20 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for c in cfg5 cfg2 cfg3 cfg4; do
  timeout -k 10 300 python3 $R/bench.py --config $c 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json || exit 1
done
B="python3 $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-impact-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats/raw -- $B > $O/stats/run.log 2>&1 || exit 1
cp $(find $O/stats/raw -name "*kernel_stats.csv" | head -1) $O/stats/cfg5_kernel_stats.csv
cp $(find $O/stats/raw -name "*kernel_trace.csv" | head -1) $O/stats/cfg5_kernel_trace.csv
rm -rf $O/stats/raw
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch/raw -- $B > $O/pmc_fetch/run.log 2>&1 || exit 1
cp $(find $O/pmc_fetch/raw -name "*counter_collection.csv" | head -1) $O/pmc_fetch/cfg5_counter_collection.csv; rm -rf $O/pmc_fetch/raw
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write/raw -- $B > $O/pmc_write/run.log 2>&1 || exit 1
cp $(find $O/pmc_write/raw -name "*counter_collection.csv" | head -1) $O/pmc_write/cfg5_counter_collection.csv; rm -rf $O/pmc_write/raw
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq/raw -- $B > $O/pmc_sq/run.log 2>&1 || exit 1
cp $(find $O/pmc_sq/raw -name "*counter_collection.csv" | head -1) $O/pmc_sq/cfg5_counter_collection.csv; rm -rf $O/pmc_sq/raw
cd $R
timeout -k 10 400 python3 tools/law_bench.py > $O/law_bench.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/law_bench.py --impacts --laws cfg5,cfg5_thin,cfg5_tile,cfg5_gen,cfg3,cfg3_k10,cfg5_q1,cfg5_q64,cfg5_q1024 >> $O/law_bench.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/e2e_bench.py > $O/e2e.txt 2>&1 || exit 1
# the widening steps' own benches
timeout -k 10 300 python3 tools/invert_bench.py > $O/invert_bench.json 2> $O/invert_bench.err || exit 1
timeout -k 10 300 python3 tools/invert_bench.py --docs 1000000 --no-cpu > $O/invert_bench_1m.json 2>> $O/invert_bench.err || exit 1
timeout -k 10 300 python3 tools/sem_bench.py > $O/sem_bench.json 2> $O/sem_bench.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/inv_raw -- python3 $R/tools/invert_bench.py --no-cpu > /dev/null 2>&1 || exit 1
cp $(find $O/inv_raw -name "*kernel_stats.csv" | head -1) $O/invert_kernel_stats.csv; rm -rf $O/inv_raw
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sem_raw -- python3 $R/tools/sem_bench.py > /dev/null 2>&1 || exit 1
cp $(find $O/sem_raw -name "*kernel_stats.csv" | head -1) $O/sem_kernel_stats.csv; rm -rf $O/sem_raw
cd $R
echo collected

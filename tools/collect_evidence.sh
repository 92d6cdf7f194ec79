#!/bin/bash
# Collects a round's evidence on the GPU box into gpurun_out/$1 (run through gpurun from the repo root); the filing script
# tools/dbg/file_evidence.py <dir> <round> then copies the summaries under profiles/<round>/ (run in the container).
#   part A: tests, bench lines of the four configs, kernel trace + stats of the kernel leg of the cfg5 bench command, L2-miss
#           traffic and SQ counters of the same command (separate --pmc passes)
#   part B: the query laws over the 1M-doc index and over the 20 x 1M-doc index (raw / packed / impact streams) with
#           FETCH_SIZE of the big-index laws, host-inclusive rates, the widening steps' own benches, 2-rank rehearsal
set -o pipefail
D=${1:-ev}
PART=${2:-AB}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$D
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() {   # prof <name> <rocprofv3 options...> -- <command...>
  local name=$1; shift
  mkdir -p $O/$name
  timeout -k 10 400 rocprofv3 "$@" > $O/$name/run.log 2>&1 || { tail -20 $O/$name/run.log; return 1; }
}
B="python3 $R/bench.py --kernel-only --steps 10 --warmup 2 --cpu-seconds 0"
if [[ $PART == *A* ]]; then
  timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider > $O/tests.txt 2>&1 || { tail -20 $O/tests.txt; exit 1; }
  tail -2 $O/tests.txt
  for c in cfg5 cfg2 cfg3 cfg4; do
    timeout -k 10 400 python3 $R/bench.py --config $c 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json || exit 1
  done
  prof stats --kernel-trace --stats --output-format csv -d $O/stats/raw -- $B || exit 1
  cp $(find $O/stats/raw -name "*kernel_stats.csv" | head -1) $O/stats/cfg5_kernel_stats.csv
  cp $(find $O/stats/raw -name "*kernel_trace.csv" | head -1) $O/stats/cfg5_kernel_trace.csv; rm -rf $O/stats/raw
  prof pmc_fetch --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch/raw -- $B || exit 1
  cp $(find $O/pmc_fetch/raw -name "*counter_collection.csv" | head -1) $O/pmc_fetch/cfg5_counter_collection.csv; rm -rf $O/pmc_fetch/raw
  prof pmc_write --pmc WRITE_SIZE --output-format csv -d $O/pmc_write/raw -- $B || exit 1
  cp $(find $O/pmc_write/raw -name "*counter_collection.csv" | head -1) $O/pmc_write/cfg5_counter_collection.csv; rm -rf $O/pmc_write/raw
  # the hbm_resident leg alone (20 x 1M docs): kernel stats and L2-miss traffic of exactly that launch
  H="python3 $R/bench.py --hbm-only --steps 10"
  prof hbm_stats --kernel-trace --stats --output-format csv -d $O/hbm_stats/raw -- $H || exit 1
  cp $(find $O/hbm_stats/raw -name "*kernel_stats.csv" | head -1) $O/hbm_stats/hbm_leg_kernel_stats.csv; rm -rf $O/hbm_stats/raw
  grep -h "hbm_resident" $O/hbm_stats/run.log | tail -1 > $O/hbm_stats/hbm_leg.json
  prof hbm_fetch --pmc FETCH_SIZE --output-format csv -d $O/hbm_fetch/raw -- $H || exit 1
  cp $(find $O/hbm_fetch/raw -name "*counter_collection.csv" | head -1) $O/hbm_fetch/hbm_leg_counter_collection.csv; rm -rf $O/hbm_fetch/raw
  prof hbm_write --pmc WRITE_SIZE --output-format csv -d $O/hbm_write/raw -- $H || exit 1
  cp $(find $O/hbm_write/raw -name "*counter_collection.csv" | head -1) $O/hbm_write/hbm_leg_counter_collection.csv; rm -rf $O/hbm_write/raw
  prof pmc_sq --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq/raw -- $B || exit 1
  cp $(find $O/pmc_sq/raw -name "*counter_collection.csv" | head -1) $O/pmc_sq/cfg5_counter_collection.csv; rm -rf $O/pmc_sq/raw
  echo "part A collected"
fi
if [[ $PART == *B* ]]; then
  cd $R
  timeout -k 10 400 python3 tools/law_bench.py > $O/law_bench.txt 2>&1 || exit 1
  for opt in "--impacts" "--prune" "--packed 1"; do
    timeout -k 10 300 python3 tools/law_bench.py $opt --laws cfg5,cfg5_thin,cfg5_tile,cfg5_gen,cfg5_top1,r1,r8,r100,cfg3,cfg3_k10,cfg5_q1,cfg5_q64,cfg5_q1024,cfg5_q2048 >> $O/law_bench.txt 2>&1 || exit 1
  done
  # the merge body and the XCD dealing, each off (same library: environment switches read at ns_ctx_create)
  for sw in "NS_MERGE=0" "NS_ORDER_MODE=0"; do
    echo "#### $sw" >> $O/law_bench.txt
    env $sw timeout -k 10 300 python3 tools/law_bench.py --laws cfg5,cfg5_thin,cfg5_tile,cfg5_gen,cfg5_t2_gen,r8r20,r40r60,r8r300,cfg3,cfg5_q2048 >> $O/law_bench.txt 2>&1 || exit 1
  done
  # every posting scored in place (ns_ctx_share_scores(0)): the laws that share term scores by default
  echo "#### NS_SHARE=0" >> $O/law_bench.txt
  NS_SHARE=0 timeout -k 10 300 python3 tools/law_bench.py --laws r1,r8,cfg5,cfg5_seed7,cfg5_top1,cfg5_thin,cfg5_tile,cfg5_gen,cfg5_2hot_gen,cfg5_t2_gen,cfg3,cfg3_k10,cfg5_q4096,cfg5_seed99_q32768,hot5_k10 >> $O/law_bench.txt 2>&1 || exit 1
  BIG="--segments 20 --qscale 0.125"
  BL=cfg5,cfg5_thin,cfg5_tile,cfg5_gen,scan_once,r1,r8,r100,r1000,cfg3_k10,cfg5_top1
  for opt in "" "--packed 1" "--prune" "--impacts"; do
    timeout -k 10 400 python3 tools/law_bench.py $BIG $opt --laws $BL >> $O/law_big20.txt 2>&1 || exit 1
  done
  echo "#### NS_ORDER_MODE=0" >> $O/law_big20.txt
  NS_ORDER_MODE=0 timeout -k 10 400 python3 tools/law_bench.py $BIG --laws cfg5,cfg5_thin,cfg5_tile,cfg5_gen >> $O/law_big20.txt 2>&1 || exit 1
  cd /tmp
  for tag in raw nodeal; do
    case $tag in raw) opt=""; export NS_ORDER_MODE=1;; nodeal) opt=""; export NS_ORDER_MODE=0;; esac
    prof big_fetch_$tag --pmc FETCH_SIZE --output-format csv -d $O/big_fetch_$tag/raw -- python3 $R/tools/law_bench.py $BIG $opt --laws cfg5,cfg5_thin,cfg5_gen,cfg5_tile,r8 --reps 3 || exit 1
    cp $(find $O/big_fetch_$tag/raw -name "*counter_collection.csv" | head -1) $O/big_fetch_$tag/counter_collection.csv; rm -rf $O/big_fetch_$tag/raw
  done
  unset NS_ORDER_MODE
  cd $R
  timeout -k 10 400 python3 tools/e2e_bench.py > $O/e2e.txt 2>&1 || exit 1
  # the C++ facade timed from inside the process: query TEXT in -> hits in host memory out
  IDX=/tmp/ns_facade_idx
  ./nextsearch-api_amd/ns_tool gen-index $IDX 1 1000000 > /dev/null
  python3 -c "
import sys; sys.path.insert(0,'nextsearch-api_amd'); import workloads
open('/tmp/cfg5.txt','w').write('\n'.join(workloads.cfg5_queries())+'\n')
open('/tmp/cfg3.txt','w').write('\n'.join(workloads.cfg3_queries())+'\n')"
  for rep in 1 2 3; do timeout -k 10 120 ./nextsearch-api_amd/ns_tool facade-bench $IDX /tmp/cfg5.txt 10 11 0 >> $O/facade_bench.txt || exit 1; done
  timeout -k 10 120 ./nextsearch-api_amd/ns_tool facade-bench $IDX /tmp/cfg3.txt 100 7 0 >> $O/facade_bench.txt || exit 1
  for sb in 2048 4096 8192 100000; do echo "NS_SUBBATCH=$sb" >> $O/facade_bench.txt; NS_SUBBATCH=$sb timeout -k 10 120 ./nextsearch-api_amd/ns_tool facade-bench $IDX /tmp/cfg5.txt 10 11 0 >> $O/facade_bench.txt || exit 1; done
  timeout -k 10 300 python3 tools/invert_bench.py > $O/invert_bench.json 2> $O/invert_bench.err || exit 1
  timeout -k 10 300 python3 tools/invert_bench.py --docs 1000000 --no-cpu > $O/invert_bench_1m.json 2>> $O/invert_bench.err || exit 1
  timeout -k 10 300 python3 tools/sem_bench.py > $O/sem_bench.json 2> $O/sem_bench.err || exit 1
  cd /tmp
  prof inv_stats --kernel-trace --stats --output-format csv -d $O/inv_stats/raw -- python3 $R/tools/invert_bench.py --no-cpu || exit 1
  cp $(find $O/inv_stats/raw -name "*kernel_stats.csv" | head -1) $O/invert_kernel_stats.csv; rm -rf $O/inv_stats
  prof sem_stats --kernel-trace --stats --output-format csv -d $O/sem_stats/raw -- python3 $R/tools/sem_bench.py || exit 1
  cp $(find $O/sem_stats/raw -name "*kernel_stats.csv" | head -1) $O/sem_kernel_stats.csv; rm -rf $O/sem_stats
  for sc in strong weak; do
    timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 $R/bench.py --gpus 2 --backend gloo --scaling $sc --steps 10 --warmup 2 2> $O/n2_$sc.err | tail -1 > $O/n2_$sc.json || exit 1
  done
  echo "part B collected"
fi

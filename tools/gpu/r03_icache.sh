#!/bin/bash
# instruction-cache counters of the scoring kernel: per class alone and in the mix, merge body on / off
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_icache}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name env law
  env $2 NS_RELOAD_WARMUP=0 timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/raw_$1 -- python3 $R/tools/law_bench.py --laws $3 --reps 3 > $O/$1.log 2>&1 || { echo "pass $1 failed"; tail -5 $O/$1.log; return 0; }
  cp $(find $O/raw_$1 -name "*counter_collection.csv" | head -1) $O/$1.csv; rm -rf $O/raw_$1
}
run thin NS_MERGE=0 cfg5_thin
run gen NS_MERGE=0 cfg5_gen
run tile NS_MERGE=0 cfg5_tile
run mix_m0 NS_MERGE=0 cfg5
run mix_m1 NS_MERGE=1 cfg5
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/*.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_uscore" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    d = {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()}
    miss = d.get("SQC_ICACHE_MISSES", 0) / max(d.get("SQC_ICACHE_REQ", 1), 1)
    print(f.split("/")[-1], d, "miss rate %.4f" % miss)
PY

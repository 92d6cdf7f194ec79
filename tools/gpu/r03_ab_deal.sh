#!/bin/bash
# A/B on one box: base library vs working tree with the dealing off / on, on the 1M-doc index and the 20-segment index,
# plus FETCH_SIZE of the 20-segment cfg5 launch both ways and the facade bench.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_ab_deal}
mkdir -p $O
cd $R
echo "#### kernel change alone (dealing off)"
NS_ORDER_MODE=0 bash tools/gpu/ab.sh ${1:-r03_ab_deal}_nodeal cfg5_thin,cfg5_tile,cfg5_gen,cfg5_2hot_gen,cfg5,cfg3,cfg5_q2048 || exit 1
echo "#### defaults (dealing on)"
bash tools/gpu/ab.sh ${1:-r03_ab_deal} cfg5_thin,cfg5_tile,cfg5_gen,cfg5_2hot_gen,cfg5,cfg3,cfg5_q2048,cfg5_q4096 || exit 1
echo "#### 20 x 1M docs: dealing off, on (c3), on (c4)"
for m in "0 3" "1 3" "1 4" "0 3" "1 3"; do
  set -- $m
  echo "== mode $1 coarse $2" | tee -a $O/big20.txt
  NS_ORDER_MODE=$1 NS_ORDER_COARSE=$2 timeout -k 10 400 python3 tools/law_bench.py --segments 20 --qscale 0.125 --laws cfg5,cfg5_thin,cfg5_gen,cfg5_tile --reps 5 2>&1 | grep -v "^variant\|^  *law" | tee -a $O/big20.txt || exit 1
done
cd /tmp && export TMPDIR=/tmp
fetch() {
  NS_RELOAD_WARMUP=0 NS_ORDER_MODE=$2 NS_ORDER_COARSE=$3 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/raw_$1 -- python3 $R/tools/law_bench.py --segments 20 --qscale 0.125 --laws cfg5 --reps 2 > $O/fetch_$1.log 2>&1 || { echo "fetch $1 failed"; tail -5 $O/fetch_$1.log; return 0; }
  cp $(find $O/raw_$1 -name "*counter_collection.csv" | head -1) $O/fetch_$1.csv; rm -rf $O/raw_$1
}
fetch off 0 3
fetch c3 1 3
fetch c4 1 4
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/fetch_*.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_uscore" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f.split("/")[-1], "L2-miss GB per launch (2 x FETCH_SIZE x 1024):", {k: round(2*1024*v[0] / max(v[1], 1) / 1e9, 2) for k, v in acc.items()}, "dispatches", max((v[1] for v in acc.values()), default=0))
PY

#!/bin/bash
# Round 3, experiment: launch order inside coarse run-time classes by (segment, doc range, largest list) [+ XCD dealing].
# Same box, same library; the mode comes from the environment (NS_ORDER_MODE / NS_ORDER_COARSE).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_order}
mkdir -p $O
cd $R
LAWS20=cfg5,cfg5_thin,cfg5_gen,cfg5_tile
LAWS1=cfg5,cfg5_gen,cfg3,cfg5_q2048
run() {  # name mode coarse
  echo "== $1 big20" >> $O/order.txt
  NS_ORDER_MODE=$2 NS_ORDER_COARSE=$3 timeout -k 10 400 python3 tools/law_bench.py --segments 20 --qscale 0.125 --laws $LAWS20 --reps 5 2>&1 | grep -v "^variant\|^  *law" >> $O/order.txt || exit 1
  echo "== $1 1m" >> $O/order.txt
  NS_ORDER_MODE=$2 NS_ORDER_COARSE=$3 timeout -k 10 300 python3 tools/law_bench.py --laws $LAWS1 --reps 8 2>&1 | grep -v "^variant\|^  *law" >> $O/order.txt || exit 1
}
run base 0 3
run key_c3 2 3
run xcd_c3 1 3
run xcd_c5 1 5
run xcd_c7 1 7
run key_c5 2 5
run base2 0 3
cd /tmp && export TMPDIR=/tmp
fetch() {  # name mode coarse
  NS_ORDER_MODE=$2 NS_ORDER_COARSE=$3 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/raw_$1 -- python3 $R/tools/law_bench.py --segments 20 --qscale 0.125 --laws cfg5 --reps 2 > $O/fetch_$1.log 2>&1 || { echo "fetch $1 failed"; tail -5 $O/fetch_$1.log; return 0; }
  cp $(find $O/raw_$1 -name "*counter_collection.csv" | head -1) $O/fetch_$1.csv; rm -rf $O/raw_$1
}
fetch base 0 3
fetch xcd_c3 1 3
fetch xcd_c5 1 5
fetch key_c3 2 3
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/fetch_*.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_uscore" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f.split("/")[-1], {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()}, "x2x1024 GB:", {k: round(2*1024*v[0] / max(v[1], 1) / 1e9, 2) for k, v in acc.items()}, "dispatches", max((v[1] for v in acc.values()), default=0))
PY
cat $O/order.txt

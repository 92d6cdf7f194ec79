#!/bin/bash
# Instruction counts of the scoring kernel per cfg5 class (one --pmc pass per law), then the event counts of the counting build.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_insts}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for law in ${2:-cfg5_thin cfg5_tile cfg5_gen cfg5}; do
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/raw_$law -- python3 $R/tools/law_bench.py --laws $law --reps 3 > $O/$law.log 2>&1 || { echo "pass $law failed"; tail -5 $O/$law.log; exit 1; }
  cp $(find $O/raw_$law -name "*counter_collection.csv" | head -1) $O/$law.csv; rm -rf $O/raw_$law
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/*.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_uscore" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f.split("/")[-1], {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()}, "dispatches", max((v[1] for v in acc.values()), default=0))
PY
cd $R && bash tools/gpu/count.sh ${1:-r03_insts} cfg5_thin,cfg5_tile,cfg5_gen,cfg5

#!/bin/bash
# Memory-side counters (vector L1, address translation, L2, fabric requests) of the scoring kernel on one query law.
#   pmc_mem.sh OUTDIR LAW [extra law_bench args]
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmcm}
LAW=${2:-r1r2r3r4r5}
EXTRA=${3:-}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/raw_$name -- python3 $R/tools/law_bench.py --laws $LAW --reps 3 $EXTRA > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/$name.log; return 0; }
  cp $(find $O/raw_$name -name "*counter_collection.csv" | head -1) $O/$name.csv; rm -rf $O/raw_$name
}
pass tcp1 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
pass tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tcp3 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
pass tcc2 TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_LEVEL_sum
pass fetch FETCH_SIZE
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/*.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_uscore" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f.split("/")[-1], {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()}, "dispatches", max((v[1] for v in acc.values()), default=0))
PY

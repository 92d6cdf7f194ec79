#!/bin/bash
# SQ / LDS / instruction-cache counters of the scoring kernel on one query law (tools/law_bench.py), in separate --pmc passes.
#   pmc_law.sh OUTDIR LAW [extra law_bench args]
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc}
LAW=${2:-r1r2r3r4r5}
EXTRA=${3:-}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/raw_$name -- python3 $R/tools/law_bench.py --laws $LAW --reps 3 $EXTRA > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/$name.log; return 0; }
  cp $(find $O/raw_$name -name "*counter_collection.csv" | head -1) $O/$name.csv; rm -rf $O/raw_$name
}
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 SQ_LEVEL_WAVES
pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_CYCLES
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/*.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_uscore" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f.split("/")[-1], {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()}, "dispatches", max((v[1] for v in acc.values()), default=0))
PY

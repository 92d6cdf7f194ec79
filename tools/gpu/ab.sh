#!/bin/bash
# Same-box A/B of builds of the HIP library: base (tools/dbg/build_ab.sh) vs the working tree's ("new"), alternating;
# VARIANTS="base new x y" adds libnextsearch_hip_x.so / _y.so (built by hand next to them) to the rotation.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-ab}
LAWS=${2:-r1,cfg5_thin,cfg5_tile,cfg5_gen,cfg5,cfg3,cfg5_q2048}
mkdir -p $O
cd $R/nextsearch-api_amd && cp libnextsearch_hip.so libnextsearch_hip_new.so
# the working tree's library is put back whatever happens (a failure mid-loop must not leave the base revision installed)
trap 'cp $R/nextsearch-api_amd/libnextsearch_hip_new.so $R/nextsearch-api_amd/libnextsearch_hip.so' EXIT
cd $R
for rep in 1 2 3; do
  for v in ${VARIANTS:-base new}; do
    cp nextsearch-api_amd/libnextsearch_hip_$v.so nextsearch-api_amd/libnextsearch_hip.so
    echo "== $v (rep $rep)" >> $O/ab.txt
    timeout -k 10 300 python3 tools/law_bench.py --laws $LAWS --reps 8 2>&1 | grep -v "^variant\|^  *law" >> $O/ab.txt || exit 1
  done
done
python3 - <<PY
import re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
v = None
for ln in open("$O/ab.txt"):
    if ln.startswith("=="):
        v = ln.split()[1]; continue
    f = ln.split()
    if len(f) >= 5:
        acc[f[0]][v].append(float(f[4]))
for law, d in acc.items():
    med = {v: sorted(x)[len(x) // 2] for v, x in d.items()}
    b = med["base"]
    print(f"{law:>14}  " + "  ".join(f"{v} {m:.3f} ms ({100 * (m / b - 1):+.1f} %)" for v, m in med.items()) + "   " + str(dict(d)))
PY

#!/bin/bash
# Same-box A/B of two source trees: _base/ (git archive of a revision, built with make) vs the working tree.
#   ab_tree.sh OUTDIR LAWS [extra law_bench args]
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-abt}
LAWS=${2:-cfg5_tile,cfg5,cfg3}
EXTRA=${3:-}
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for v in base new; do
    T=$R; [ $v = base ] && T=$R/_base
    echo "== $v (rep $rep)" >> $O/ab.txt
    timeout -k 10 300 python3 $T/tools/law_bench.py --laws $LAWS --reps 8 $EXTRA 2>&1 | grep -v "^variant\|^  *law" >> $O/ab.txt || exit 1
  done
done
python3 - <<PY
import collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
v = None
for ln in open("$O/ab.txt"):
    if ln.startswith("=="):
        v = ln.split()[1]; continue
    f = ln.split()
    if len(f) >= 5:
        try: acc[f[0]][v].append(float(f[4]))
        except ValueError: pass
for law, d in acc.items():
    b = sorted(d["base"])[len(d["base"]) // 2]; n = sorted(d["new"])[len(d["new"]) // 2]
    print(f"{law:>14}  base {b:.3f} ms  new {n:.3f} ms  {100 * (n / b - 1):+.1f} %   (base {d['base']}  new {d['new']})")
PY

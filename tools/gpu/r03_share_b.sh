#!/bin/bash
# shared term scores: the 20-segment (HBM-resident) law with sharing off / forced, then bench.py (value leg, in-place leg)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_share_b}
mkdir -p $O
cd $R
for rep in 1 2; do
  for m in 0 2; do
    echo "== share $m (rep $rep)" >> $O/big.txt
    timeout -k 10 500 python3 tools/law_bench.py --share $m --segments 20 --qscale 0.125 --laws cfg5,cfg5_thin,cfg5_gen,cfg5_tile --reps 4 2>&1 | grep -v "^variant\|^  *law" >> $O/big.txt || exit 1
  done
done
cat $O/big.txt
for m in 1 0; do
  timeout -k 10 600 python3 bench.py --share $m --no-hbm-leg --cpu-seconds 0 > $O/bench_share$m.json 2> $O/bench_share$m.err || { tail -5 $O/bench_share$m.err; exit 1; }
done
python3 - <<PY
import json
for m in (1, 0):
    d = json.load(open("$O/bench_share%d.json" % m))
    print("share", m, "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "kernel_ms", round(d["roofline"]["kernel_ms"], 3), "frac", round(d["roofline"]["frac"], 4),
          "in_place", d.get("in_place", {}).get("kernel_ms"), "kernel_only", round(d["kernel_only"]["ms_per_step"], 3))
PY

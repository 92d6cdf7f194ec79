#!/bin/bash
# kernel timeline of the pipelined bench loop with shared term scores (who waits for whom)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_share_trace}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/raw$m -- python3 $R/bench.py --share $m --no-hbm-leg --no-impact-leg --cpu-seconds 0 --steps 12 --warmup 3 > $O/bench$m.json 2> $O/bench$m.err || { tail -5 $O/bench$m.err; exit 1; }
  cp $(find $O/raw$m -name "*kernel_trace.csv" | head -1) $O/trace$m.csv; rm -rf $O/raw$m
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/trace$m.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# the last 40 kernels of the pipelined loop (before the tail legs): print name, queue, start, end in us
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_uscore", "k_share_scores", "k_merge", "k_pull"))]
tail = sel[-150:-100] if len(sel) > 160 else sel[-50:]
print("share", $m, "kernels", len(sel))
for r in tail:
    print(f'{r["Kernel_Name"][:28]:28s} q{r["Queue_Id"]:>3} {(int(r["Start_Timestamp"]) - t0) / 1e3:12.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:12.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f}')
PY
done

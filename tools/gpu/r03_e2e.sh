#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_e2e}
mkdir -p $O
cd $R
timeout -k 10 500 python3 tools/e2e_bench.py > $O/e2e.txt 2>&1 || { tail -20 $O/e2e.txt; exit 1; }
grep -n "host threads\|batches of\|rep [0-9]:" $O/e2e.txt | cut -c1-260
for c in cfg5 cfg3 cfg4 cfg2; do
  timeout -k 10 600 python3 bench.py --config $c 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json || { tail -20 $O/bench_$c.err; exit 1; }
done
python3 - <<PY
import json
for c in ("cfg5", "cfg3", "cfg4", "cfg2"):
    d = json.load(open("$O/bench_%s.json" % c))
    print(c, "value %.0f q/s  ms/step %.3f  frac %.4f kernel_ms %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"]), "hbm", d.get("hbm_resident", {}).get("roofline", {}).get("frac"))
PY
for sc in strong weak; do
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 $R/bench.py --gpus 2 --backend gloo --scaling $sc --steps 10 --warmup 2 2> $O/n2_$sc.err | tail -1 > $O/n2_$sc.json || { tail -20 $O/n2_$sc.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/n2_$sc.json')); print('n2 $sc', d['value'], d['ms_per_step'])"
done

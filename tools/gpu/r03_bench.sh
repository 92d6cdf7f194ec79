#!/bin/bash
# full GPU suite, the default bench line, the facade bench (text in -> hits out, in-process), pruning on the single-term laws
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_bench}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest $R/tests -m gpu -x -q -p no:cacheprovider --durations=5 > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -4 $O/tests.txt
cd $R
timeout -k 10 600 python3 bench.py 2>$O/bench.err | tail -1 > $O/bench.json || { tail -20 $O/bench.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print("value %.0f q/s  ms/step %.3f  roofline frac %.3f kernel_ms %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"]))
for k in ("hbm_resident", "impact_stream", "pruned", "cpu_baseline"):
    if k in d: print(k, json.dumps(d[k])[:600])
PY
IDX=/tmp/ns_facade_idx
./nextsearch-api_amd/ns_tool gen-index $IDX 1 1000000 > /dev/null
python3 -c "
import sys; sys.path.insert(0,'nextsearch-api_amd'); import workloads
open('/tmp/cfg5.txt','w').write('\n'.join(workloads.cfg5_queries())+'\n')
open('/tmp/cfg3.txt','w').write('\n'.join(workloads.cfg3_queries())+'\n')"
for rep in 1 2; do
  timeout -k 10 120 ./nextsearch-api_amd/ns_tool facade-bench $IDX /tmp/cfg5.txt 10 9 0 | tee -a $O/facade.txt
done
timeout -k 10 120 ./nextsearch-api_amd/ns_tool facade-bench $IDX /tmp/cfg3.txt 100 5 0 | tee -a $O/facade.txt

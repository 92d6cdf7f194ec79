#!/bin/bash
# final check of the round: smoke(), then evidence part A (full GPU suite, four bench lines, kernel stats and PMC passes)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" || exit 1
bash tools/collect_evidence.sh ${1:-ev_r03_final} A

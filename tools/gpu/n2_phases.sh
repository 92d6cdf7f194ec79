#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 $R/tools/dbg/n2_phases.py gloo 2>&1 | grep -v "^W\|^\*\*\*\|Setting OMP" | tail -8

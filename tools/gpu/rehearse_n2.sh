#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on the one-GPU box: two ranks share the device; gloo carries the collective
# (the nccl == RCCL configuration needs one GPU per rank and is what the driver's scaling run measures).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-n2}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sc in strong weak; do
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 $R/bench.py --gpus 2 --backend gloo --scaling $sc --steps 10 --warmup 2 > $O/n2_$sc.json 2> $O/n2_$sc.err || { tail -30 $O/n2_$sc.err; exit 1; }
  tail -1 $O/n2_$sc.json
done

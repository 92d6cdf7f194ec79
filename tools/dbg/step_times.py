#!/usr/bin/env python3
"""Per-step completion times of the pipelined loop bench.py times (GPU box): where the slow steps are.
    python3 tools/dbg/step_times.py [share mode] [steps]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
import numpy as np
import nsbind, workloads
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tmp = tempfile.TemporaryDirectory(); idx = os.path.join(tmp.name, "i")
nsbind.gen_index(idx, 1, 1_000_000, 65536, 1337, False)
eng = nsbind.Engine(idx, 0)
eng.share_scores(mode)
L = nsbind.hip_lib()
rot = [eng.build_refs(workloads.cfg5_queries(16384, 2005 + 104729 * i))[:2] for i in range(4)]
L.ns_ctx_set_overlap(eng.ctx, 1)
Q, K = 16384, 10
out = [(np.empty((Q, K), dtype=nsbind.HIT_DTYPE), np.empty(Q, np.uint32), np.empty(Q, np.uint64)) for _ in range(2)]
for rnd in range(3):
    t0 = time.perf_counter(); ts = []
    for res in nsbind.pipelined_search(eng.ctx, [rot[i % 4] for i in range(steps)], K, 0, out=out, depth=3):
        ts.append(time.perf_counter() - t0)
    d = [ts[0]] + [b - a for a, b in zip(ts, ts[1:])]
    print(f"share {mode} round {rnd}: total {ts[-1] * 1e3:.2f} ms / {steps} = {ts[-1] / steps * 1e3:.3f}; first step at {d[0] * 1e3:.2f} ms; step intervals (ms): " + " ".join(f"{x * 1e3:.2f}" for x in d[1:]))
eng.close()

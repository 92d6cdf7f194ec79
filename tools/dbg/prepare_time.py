#!/usr/bin/env python3
"""Host time of ns_batch_prepare alone (GPU box): cfg5 batches of several sizes, sharing off / default rule / forced.
    python3 tools/dbg/prepare_time.py [Q,Q,...]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
import nsbind, workloads
tmp = tempfile.TemporaryDirectory(); idx = os.path.join(tmp.name, "i")
nsbind.gen_index(idx, 1, 1_000_000, 65536, 1337, False)
eng = nsbind.Engine(idx, 0)
for Q in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16384,4096,2048").split(",")]:
    sets = [eng.build_refs(workloads.cfg5_queries(Q, 2005 + 104729 * i))[:2] for i in range(4)]
    for mode in (0, 1, 2):
        eng.share_scores(mode)
        for rep in range(2):
            t0 = time.perf_counter()
            n = 0
            for i in range(12):
                qd, refs = sets[i % 4]
                b = nsbind.prepare_raw(eng.ctx, qd, refs, 10, 0)
                inf = b.info()
                b.close()
                n += 1
            dt = (time.perf_counter() - t0) / n * 1e3
        print(f"Q={Q:6d} share={mode}: prepare + destroy {dt:.3f} ms per batch (shared lists {inf.shared_lists}, flags {inf.flags:#x})")
eng.close()

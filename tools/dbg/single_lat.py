#!/usr/bin/env python3
"""One query at a time through prepare/run/fetch/destroy (for a kernel + memory-copy trace of the
single-query path).  Diagnostic; GPU box only."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
import nsbind, workloads
tmp = tempfile.TemporaryDirectory(); idx = os.path.join(tmp.name, "i")
nsbind.gen_index(idx, 1, 1_000_000, 65536, 1337, False)
eng = nsbind.Engine(idx, 0)
eng.set_cache(False)
qs = workloads.cfg5_queries()[:200]
for rep in range(3):
    acc = [0.0] * 4
    for q in qs:
        t1 = time.perf_counter()
        b = eng.prepare([q], 10); t2 = time.perf_counter()
        b.run(False); b.sync(); t3 = time.perf_counter()
        b.fetch(); t4 = time.perf_counter(); b.close(); t5 = time.perf_counter()
        for i, d in enumerate((t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[i] += d
    print("rep %d us: prepare %.0f | run+sync %.0f | fetch %.0f | destroy %.0f" % ((rep,) + tuple(1e6 * a / len(qs) for a in acc)))
    t0 = time.perf_counter()
    for q in qs:
        eng.search_json(q, 10)
    print("rep %d search_json %.0f us" % (rep, 1e6 * (time.perf_counter() - t0) / len(qs)))
eng.close()

#!/usr/bin/env python3
"""Diagnostic: do two scoring launches on the ctx's two streams overlap?  (GPU box only)"""
import ctypes as C, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
import numpy as np
import nsbind, workloads
L = nsbind.hip_lib()
tmp = tempfile.TemporaryDirectory(); idx = os.path.join(tmp.name, "i")
nsbind.gen_index(idx, 1, 1_000_000, 65536, 1337, False)
os.environ["NS_RELOAD_WARMUP"] = "0"
eng = nsbind.Engine(idx, 0)
for nq in (16384, 2048):
    qs = workloads.cfg5_queries(nq)
    qd, refs, _ = eng.build_refs(qs)
    for overlap in (0, 1):
        L.ns_ctx_set_overlap(eng.ctx, overlap)
        b1 = nsbind.prepare_raw(eng.ctx, qd, refs, 10); b2 = nsbind.prepare_raw(eng.ctx, qd, refs, 10)
        for _ in range(4):
            b1.run(); b2.run()
        b1.sync(); b2.sync()
        n = 12
        t0 = time.perf_counter()
        for _ in range(n):
            b1.run(timed=True); b2.run(timed=True)
        b1.sync(); b2.sync()
        dt = time.perf_counter() - t0
        i1, i2 = b1.info(), b2.info()
        g = C.c_float(); rc = L.ns_batch_gap_ms(b1.h, b2.h, C.byref(g))
        print(f"{nq} queries, overlap {overlap}: {1e3 * dt / (2 * n):.3f} ms per launch (wall); HIP-event kernel time {i1.sum_total_ms / i1.timed_runs:.3f} / {i2.sum_total_ms / i2.timed_runs:.3f} ms; "
              f"gap from the first batch's first timed run end to the second's first start {g.value:.3f} ms (rc {rc}); streams {b1.stream:#x} {b2.stream:#x}")
        b1.close(); b2.close()
eng.close()

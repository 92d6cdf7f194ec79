#!/usr/bin/env python3
"""End-to-end (PCIe- and host-inclusive) rates of the same batch that bench.py times device-only.
Diagnostic; prints one line per stage.  GPU box only."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
import nsbind, workloads
tmp = tempfile.TemporaryDirectory(); idx = os.path.join(tmp.name, "i")
nsbind.gen_index(idx, 1, 1_000_000, 65536, 1337, False)
t0 = time.perf_counter(); eng = nsbind.Engine(idx, 0); t1 = time.perf_counter(); eng.set_cache(False)   # time searches, not the result cache (src/api_engine.cpp:380-385)
print(f"Engine.reload (load 83 MB index from disk + upload + norms): {t1 - t0:.3f} s")
qs = workloads.cfg5_queries(); Q = len(qs)
eng.search_batch(qs[:64], 10)
for rep in range(3):
    t0 = time.perf_counter(); qd, refs, usable = eng.build_refs(qs); t1 = time.perf_counter()
    rc, hits, nhits, found = nsbind.search_batch_raw(eng.ctx, qd, refs, 10); t2 = time.perf_counter()
    b = eng.prepare(qs, 10); t3 = time.perf_counter()
    b.run(False); b.sync(); t4 = time.perf_counter()
    b.fetch(); t5 = time.perf_counter(); b.close()
    t6 = time.perf_counter(); eng.search_batch(qs, 10); t7 = time.perf_counter()
    print(f"rep {rep}: query prep (tokenise+lexicon+idf) {1e3*(t1-t0):.1f} ms | ns_search_batch (host->host) {1e3*(t2-t1):.1f} ms = {Q/(t2-t1):.0f} q/s | "
          f"prepare(incl. query prep) {1e3*(t3-t2):.1f} ms, run {1e3*(t4-t3):.2f} ms, fetch {1e3*(t5-t4):.2f} ms | facade search_batch {1e3*(t7-t6):.1f} ms = {Q/(t7-t6):.0f} q/s")
# pipelined host -> host at the C-ABI (what bench.py reports as `value`): descriptors in host memory in, results in host
# memory out, two batches in flight on the one ctx: prepare(i+1) || run(i) || fetch(i-1)
import numpy as np  # noqa: E402
L = nsbind.hip_lib()
qd, refs, usable = eng.build_refs(qs)
out = [(np.empty((Q, 10), dtype=nsbind.HIT_DTYPE), np.empty(Q, np.uint32), np.empty(Q, np.uint64)) for _ in range(2)]
for threads in (1, 2, 4, 8, 0):
    L.ns_ctx_set_host_threads(eng.ctx, threads)
    t_prep = []
    for _ in range(5):
        t0 = time.perf_counter(); b = nsbind.prepare_raw(eng.ctx, qd, refs, 10); t_prep.append(time.perf_counter() - t0); b.close()
    n = 24
    for _ in nsbind.pipelined_search(eng.ctx, [(qd, refs)] * 4, 10, out=out):
        pass
    t0 = time.perf_counter()
    for _ in nsbind.pipelined_search(eng.ctx, [(qd, refs)] * n, 10, out=out):
        pass
    dt = time.perf_counter() - t0
    print(f"host threads {threads or 'auto'}: ns_batch_prepare alone {1e3 * min(t_prep):.2f} ms (best of 5) | pipelined host->host {1e3 * dt / n:.2f} ms per batch = {Q * n / dt:.0f} q/s")
L.ns_ctx_set_host_threads(eng.ctx, 0)
# where a pipelined step's time goes: the kernels' own HIP-event times inside the pipelined loop (one stream, then two)
import ctypes as C  # noqa: E402
from collections import deque  # noqa: E402
for overlap in (0, 1):
    L.ns_ctx_set_overlap(eng.ctx, overlap)
    ks, ts, gaps = [], [], []
    flight = deque()
    t0 = time.perf_counter()
    for i in range(24):
        b = nsbind.prepare_raw(eng.ctx, qd, refs, 10); b.run(timed=True, fetch=True); flight.append(b)
        if len(flight) >= 4:                      # keep the retired batch's successor alive: the gap needs both
            old = flight.popleft(); old.fetch_into(*out[0]); inf = old.info()
            g = C.c_float()
            if L.ns_batch_gap_ms(old.h, flight[0].h, C.byref(g)) == 0:
                gaps.append(g.value)
            ks.append(inf.last_score_kernel_ms); ts.append(inf.last_total_ms); old.close()
    while flight:
        old = flight.popleft(); old.fetch_into(*out[0]); old.close()
    dt = time.perf_counter() - t0
    print(f"pipelined, 4 in flight, overlap {overlap}: {1e3 * dt / 24:.3f} ms per batch; scoring kernel by HIP events {sum(ks[4:]) / len(ks[4:]):.3f} ms, all kernels {sum(ts[4:]) / len(ts[4:]):.3f} ms, "
          f"device gap between consecutive batches {sum(gaps[4:]) / max(len(gaps[4:]), 1):.3f} ms (min {min(gaps[4:]):.3f}, max {max(gaps[4:]):.3f})")
L.ns_ctx_set_overlap(eng.ctx, 0)
# batches alternating between two streams (ns_ctx_set_overlap): the tail of batch i overlaps the head of batch i+1
for nq in (16384, 4096, 2048, 1024, 256):
    qd_s, refs_s, _ = eng.build_refs(qs[:nq])
    for overlap, depth in ((0, 2), (0, 3), (1, 2), (1, 3), (1, 4)):
        L.ns_ctx_set_overlap(eng.ctx, overlap)
        n = max(24, 65536 // nq)
        for _ in nsbind.pipelined_search(eng.ctx, [(qd_s, refs_s)] * 6, 10, out=out, depth=depth):
            pass
        t0 = time.perf_counter()
        for _ in nsbind.pipelined_search(eng.ctx, [(qd_s, refs_s)] * n, 10, out=out, depth=depth):
            pass
        dt = time.perf_counter() - t0
        print(f"batches of {nq} queries, overlap {overlap}, {depth} in flight: pipelined host->host {1e3 * dt / n:.3f} ms per batch = {nq * n / dt:.0f} q/s")
L.ns_ctx_set_overlap(eng.ctx, 0)
# query TEXT in -> /api/search JSON bodies out (result assembly incl. metadata.csv decoration when the file is there)
import workloads as _w  # noqa: E402
with open(os.path.join(idx, "metadata.csv"), "wb") as f:
    f.write(_w.metadata_csv(1_000_000, 11))
eng.close()
t0 = time.perf_counter(); eng = nsbind.Engine(idx, 0); t1 = time.perf_counter(); eng.set_cache(False)   # time searches, not the result cache (src/api_engine.cpp:380-385)
print(f"Engine.reload with a {os.path.getsize(os.path.join(idx, 'metadata.csv')) >> 20} MB metadata.csv: {t1 - t0:.3f} s")
for what in ("first 500 requests after reload (one-time costs: kernel code load, pinned staging, block pool)", "steady state"):
    t0 = time.perf_counter()
    for q in qs[:500]:
        eng.search_json(q, 10)
    t1 = time.perf_counter()
    print(f"one query at a time, Engine::search(query, 10) -> decorated JSON text, {what}: {1e6 * (t1 - t0) / 500:.0f} us per query (mean of 500)")
acc = [0.0] * 5
for q in qs[:500]:
    t0 = time.perf_counter(); qd, refs, usable = eng.build_refs([q]); t1 = time.perf_counter()
    b = eng.prepare([q], 10); t2 = time.perf_counter()
    b.run(False); b.sync(); t3 = time.perf_counter()
    b.fetch(); t4 = time.perf_counter(); b.close(); t5 = time.perf_counter()
    for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[i] += d
print("one query at a time, by stage (us, mean of 500, ctypes call overhead included): query prep %.0f | prepare (incl. query prep) %.0f | "
      "run+sync %.0f | fetch %.0f | destroy %.0f" % tuple(1e6 * a / 500 for a in acc))
eng.search_batch_json(qs[:64], 10)
for rep in range(3):
    t0 = time.perf_counter(); raw, offs = eng.search_batch_json(qs, 10, decode=False); t1 = time.perf_counter()
    print(f"rep {rep}: facade search_batch_json (text in, decorated JSON bodies out, {len(raw) >> 20} MB) {1e3*(t1-t0):.1f} ms = {Q/(t1-t0):.0f} q/s")
eng.close()


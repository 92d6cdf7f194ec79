#!/usr/bin/env python3
"""Diagnostic: per-work-item wall time of k_uscore vs the host's work estimate (needs the -DNS_STAMP
build copied over nextsearch-api_amd/libnextsearch_hip.so on the GPU box)."""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import nsbind, law_bench, workloads
L = nsbind.hip_lib()
tmp = tempfile.TemporaryDirectory(); idx = os.path.join(tmp.name, "i")
N = 1_000_000
nsbind.gen_index(idx, 1, N, 65536, 1337, False)
eng = nsbind.Engine(idx, 0)
laws = law_bench.laws()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
qs, k = laws[name]
b = eng.prepare(qs, k)
b.run(False); b.sync()
b.run(False); b.sync()
n = b.info().n_items
items = np.zeros((n, 8), dtype=np.uint32)
L.ns_debug_items.restype = C.c_uint
L.ns_debug_items.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
L.ns_debug_item_times.argtypes = [C.c_void_p, C.c_uint]
got = L.ns_debug_items(b.h, items.ctypes.data, n)
times = np.zeros((n, 2), dtype=np.uint64)
L.ns_debug_item_times(times.ctypes.data, n)
t0 = times[:, 0].min(); t1 = times[:, 1].max()
dur = (times[:, 1] - times[:, 0]).astype(np.float64) / 100.0   # us
span = float(t1 - t0) / 100.0
print(f"{name}: {n} items, kernel span {span:.1f} us, sum of item time {dur.sum():.0f} us = {dur.sum() / span:.0f} busy wave slots on average (6144 available)")
edges = np.linspace(0, span, 21)
st = (times[:, 0] - t0).astype(np.float64) / 100.0; en = (times[:, 1] - t0).astype(np.float64) / 100.0
for i in range(20):
    a, c = edges[i], edges[i + 1]
    busy = np.clip(np.minimum(en, c) - np.maximum(st, a), 0, None).sum() / (c - a)
    print(f"   t={a:7.0f}..{c:7.0f} us  busy slots {busy:7.0f}")
def rank_of(t):
    return workloads.WORDS.index(t) + 1 if t in workloads.WORDS else int(t[1:])
feat = []
for i in range(n):
    q = qs[items[i, 0]]
    frac = (float(items[i, 5]) - float(items[i, 4])) / N
    dfs = sorted([min(0.6 * N / rank_of(t), N) * frac for t in q.split()], reverse=True)
    cls = "tile" if items[i, 7] & 2 else ("thin" if items[i, 7] & 4 else "gen")
    feat.append((cls, dfs[0], sum(dfs[1:]), len(dfs), frac * N))
for cls in ("thin", "gen", "tile"):
    sel = [i for i in range(n) if feat[i][0] == cls]
    if not sel:
        continue
    A = np.array([[feat[i][1], feat[i][2], feat[i][3], feat[i][4], 1.0] for i in sel]); y = dur[sel]
    coef, *_ = np.linalg.lstsq(A, y, rcond=None)
    pred = A @ coef
    print(f"{cls}: {len(sel)} items, mean {y.mean():.1f} us, max {y.max():.1f} us, p99 {np.percentile(y, 99):.1f}; fit us = {coef[0]*1e3:.3f}/k driver + {coef[1]*1e3:.3f}/k foreign + {coef[2]:.2f}/term + {coef[3]*1e3:.4f}/k docs + {coef[4]:.1f}; residual rms {np.sqrt(((pred-y)**2).mean()):.1f} us")
    order = np.argsort(-y)[:5]
    for o in order:
        i = sel[o]
        print(f"      slowest: {y[o]:.1f} us  launch idx {i}  driver {feat[i][1]:.0f} foreign {feat[i][2]:.0f} terms {feat[i][3]} docs {feat[i][4]:.0f}  start {st[i]:.0f} us")
b.close()

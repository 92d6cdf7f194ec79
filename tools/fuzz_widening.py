#!/usr/bin/env python3
"""Randomised differential runs of the two widening kernels (test infrastructure; GPU box only):
index inversion against oracle/invert_oracle.py (files byte for byte), semantic top-k against the numpy
restatement of most_similar_to_vec (rows and sim bits)."""
import ctypes as C
import os
import random
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nextsearch-api_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import forward_gen  # noqa: E402
import invert_oracle  # noqa: E402
import nsbind  # noqa: E402
from test_semantic import np_most_similar, _normalised  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    tmp = tempfile.mkdtemp(prefix="ns_fz_")
    t0, n_inv, n_sem = time.time(), 0, 0
    while time.time() - t0 < seconds / 2:
        n_docs = rng.choice([1, 7, 300, 4096, 4097, 30000, 90000])
        n_terms = rng.choice([1, 2, 255, 256, 257, 5000, 65535, 65536, 65537, 200000])
        mean = rng.choice([0, 1, 5, 40, 120])
        a, b = os.path.join(tmp, f"a{n_inv}"), os.path.join(tmp, f"b{n_inv}")
        forward_gen.write_inputs(a, n_docs, n_terms, mean, rng.randint(1, 10**6), bad_ids=rng.random() < 0.5, empty_docs=rng.random() < 0.5)
        shutil.copytree(a, b)
        pa = invert_oracle.lexicon_tool(a)
        st = nsbind.invert_segment(b)
        ok = (st["pairs"], st["kept"]) == pa and all(open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read() for f in invert_oracle.output_files())
        shutil.rmtree(a); shutil.rmtree(b)
        if not ok:
            print(f"INVERSION MISMATCH docs={n_docs} terms={n_terms} mean={mean}")
            sys.exit(1)
        n_inv += 1
        if n_inv % 50 == 0:
            print(f"... {n_inv} inversions ({time.time() - t0:.0f} s)", flush=True)
    L = nsbind.hip_lib()
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    nrng = np.random.default_rng(rng.randint(1, 10**6))
    while time.time() - t0 < seconds:
        rows, dim = rng.choice([1, 63, 64, 65, 8191, 8192, 8193, 30000]), rng.choice([1, 7, 8, 9, 50, 300])
        V = _normalised(nrng, rows, dim)
        if rows > 8:
            V[rows // 2] = V[1]; V[rows - 1] = V[1]          # exact ties
        sem = C.c_void_p()
        assert L.ns_sem_upload(ctx, V.ctypes.data, rows, dim, C.byref(sem)) == 0
        n_q = rng.choice([1, 7, 8, 9, 20])
        Q = V[nrng.integers(0, rows, size=n_q)].copy()
        topk, min_sim = rng.choice([1, 3, 5, 64]), rng.choice([0.55, -2.0, 0.0, 0.99])
        bans = [sorted(set(int(x) for x in nrng.integers(0, rows, size=rng.randint(0, 3)))) for _ in range(n_q)]
        off = np.concatenate([[0], np.cumsum([len(x) for x in bans])]).astype(np.uint32)
        br = np.array([r for x in bans for r in x] or [0], dtype=np.uint32)
        orows = np.zeros((n_q, topk), dtype=np.uint32); osims = np.zeros((n_q, topk), dtype=np.float32); cnt = np.zeros(n_q, dtype=np.uint32)
        assert L.ns_sem_topk(ctx, sem, Q.ctypes.data, n_q, topk, C.c_float(min_sim), off.ctypes.data, br.ctypes.data, orows.ctypes.data, osims.ctypes.data, cnt.ctypes.data, None) == 0
        for qi in range(n_q):
            wr, ws = np_most_similar(V, Q[qi], topk, min_sim, bans[qi])
            if int(cnt[qi]) != len(wr) or orows[qi, :len(wr)].tolist() != wr or not np.array_equal(osims[qi, :len(wr)].view(np.uint32), np.asarray(ws, dtype=np.float32).view(np.uint32)):
                print(f"SEMANTIC MISMATCH rows={rows} dim={dim} topk={topk} min_sim={min_sim} q={qi}")
                sys.exit(1)
        L.ns_sem_release(ctx, sem)
        n_sem += 1
        if n_sem % 500 == 0:
            print(f"... {n_sem} similarity searches ({time.time() - t0:.0f} s)", flush=True)
    L.ns_ctx_destroy(ctx)
    print(f"fuzz: {n_inv} inversions and {n_sem} similarity searches equal to their oracles in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

"""Semantic query expansion (src/semantic_embedding.cpp).  Raw C-ABI: ns_sem_topk against a numpy
restatement of most_similar_to_vec (:104-145) with the reference's sequential fp32 dot product (:11-15)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import nsbind

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def np_dot_rows(V, q):
    """dot(q, V[r]) for every row, accumulated in index order with one fp32 rounding per multiply and per add."""
    acc = np.zeros(V.shape[0], dtype=np.float32)
    for i in range(V.shape[1]):
        acc = (acc + (np.float32(q[i]) * V[:, i]).astype(np.float32)).astype(np.float32)
    return acc


def np_most_similar(V, q, topk, min_sim, banned):
    sims = np_dot_rows(V, q)
    ok = sims >= np.float32(min_sim)
    if len(banned):
        ok[np.asarray(sorted(banned), dtype=np.int64)] = False
    rows = np.nonzero(ok)[0]
    order = sorted(rows.tolist(), key=lambda r: (-float(sims[r]), r))[:topk]
    return order, sims[order] if order else np.zeros(0, dtype=np.float32)


def _normalised(rng, rows, dim):
    V = rng.standard_normal((rows, dim)).astype(np.float32)
    n = np.sqrt((V.astype(np.float64) ** 2).sum(axis=1))
    return (V / n[:, None]).astype(np.float32)      # l2_normalize: double sum, (float)(x / n)  (:18-24)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,dim", [(20_000, 50), (8192, 300), (70_001, 64), (37, 10)])
def test_sem_topk_equals_numpy_restatement(rows, dim):
    L = nsbind.hip_lib()
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    rng = np.random.default_rng(rows + dim)
    V = _normalised(rng, rows, dim)
    # clusters, so that sims above the reference's 0.55 exist; duplicates, so that ties exist
    for c in range(0, min(rows, 400), 7):
        V[c + 1 if c + 1 < rows else c] = V[c]
        for j in range(2, 6):
            if c + j < rows:
                w = V[c] + 0.25 * j * V[c + j]
                V[c + j] = (w / np.sqrt((w.astype(np.float64) ** 2).sum())).astype(np.float32)
    sem = C.c_void_p()
    assert L.ns_sem_upload(ctx, V.ctypes.data, rows, dim, C.byref(sem)) == 0, L.ns_last_error(ctx)
    n_q = 11 if rows > 100 else 3
    qrows = rng.choice(min(rows, 400), size=n_q, replace=False)
    Q = V[qrows].copy()
    cen = Q[:3].sum(axis=0) / np.float32(3)
    Q[-1] = (cen / np.sqrt((cen.astype(np.float64) ** 2).sum())).astype(np.float32)    # a centroid query (:193-204)
    bans = [sorted({int(qrows[i]), int(qrows[(i + 1) % n_q])}) for i in range(n_q)]
    ban_off = np.concatenate([[0], np.cumsum([len(b) for b in bans])]).astype(np.uint32)
    ban_rows = np.array([r for b in bans for r in b], dtype=np.uint32)
    for topk, min_sim in ((3, 0.55), (5, 0.55), (64, -2.0), (1, 0.9999)):
        out_rows = np.zeros((n_q, topk), dtype=np.uint32)
        out_sims = np.zeros((n_q, topk), dtype=np.float32)
        cnt = np.zeros(n_q, dtype=np.uint32)
        assert L.ns_sem_topk(ctx, sem, Q.ctypes.data, n_q, topk, C.c_float(min_sim), ban_off.ctypes.data, ban_rows.ctypes.data,
                             out_rows.ctypes.data, out_sims.ctypes.data, cnt.ctypes.data, None) == 0, L.ns_last_error(ctx)
        for qi in range(n_q):
            want_rows, want_sims = np_most_similar(V, Q[qi], topk, min_sim, bans[qi])
            assert int(cnt[qi]) == len(want_rows), (rows, topk, qi)
            assert out_rows[qi, :len(want_rows)].tolist() == want_rows, (rows, topk, qi)
            assert np.array_equal(out_sims[qi, :len(want_rows)].view(np.uint32), np.asarray(want_sims, dtype=np.float32).view(np.uint32))
    # no ban lists at all
    out_rows = np.zeros((n_q, 3), dtype=np.uint32); out_sims = np.zeros((n_q, 3), dtype=np.float32); cnt = np.zeros(n_q, dtype=np.uint32)
    assert L.ns_sem_topk(ctx, sem, Q.ctypes.data, n_q, 3, C.c_float(0.55), None, None, out_rows.ctypes.data, out_sims.ctypes.data, cnt.ctypes.data, None) == 0
    for qi in range(n_q):
        want_rows, _ = np_most_similar(V, Q[qi], 3, 0.55, [])
        assert out_rows[qi, :int(cnt[qi])].tolist() == want_rows
    assert L.ns_sem_topk(ctx, sem, Q.ctypes.data, n_q, 65, C.c_float(0.5), None, None, out_rows.ctypes.data, out_sims.ctypes.data, cnt.ctypes.data, None) != 0
    assert L.ns_sem_release(ctx, sem) == 0
    L.ns_ctx_destroy(ctx)


# ------------------------------------------------------------------------------------------------------------
# End to end against the REAL reference (tests/golden/sem1.json, made by tools/gen_golden.py sem): an index with an
# embeddings file next to it -> the table the loader keeps, the weighted terms a search scores (ORDER and weight
# bits: the order is the fp32 accumulation order of the scores), and the hits.
import hashlib
import json

import workloads


def _sem_fixture(index_factory):
    with open(os.path.join(ROOT, "tests", "golden", "sem1.json")) as f:
        g = json.load(f)
    p = g["params"]
    d, _ = index_factory(p["n_segments"], p["docs_per_segment"], p["vocab"], p["seed"], p["legacy"])
    emb = workloads.embeddings_text(p["vocab"], p["emb_dim"], p["emb_seed"])
    assert hashlib.sha256(emb).hexdigest() == g["embeddings_sha256"], "the embeddings generator drifted from the golden's"
    path = os.path.join(d, "embeddings.vec")
    if not os.path.exists(path):
        with open(path, "wb") as f:
            f.write(emb)
    return g, d, path


def test_embedding_loader_equals_reference(index_factory):
    """Host only: header line, words outside the lexicons, short and odd-dimension lines, a repeated word."""
    g, d, path = _sem_fixture(index_factory)
    try:
        eng = nsbind.Engine(d, -1)
        try:
            on, rows, dim = eng.semantic_info()
            assert (int(on), rows, dim) == (g["table"]["enabled"], g["table"]["rows"], g["table"]["dim"])
            with pytest.raises(RuntimeError, match="no CPU"):      # the similarity search has no CPU path
                eng.expand("covid vaccine")
        finally:
            eng.close()
    finally:
        os.remove(path)


def test_embedding_loader_bits_equal_reference_on_odd_spellings(index_factory):
    """tests/golden/semload1.json: an embeddings file full of spellings that tell number readers apart (glued
    numbers, bare points, dangling exponents, over- and underflow, text inside a line, CR ends, a header look-alike,
    a last line without newline) and the table the REAL reference's load_from_text made of it — rows, names and the
    fp32 bits of every normalised value.  This repo's loader (own scanner, no streams) must produce the same table."""
    import base64
    from conftest import load_golden
    g = load_golden("semload1")
    p = g["params"]
    d, _ = index_factory(p["n_segments"], p["docs_per_segment"], p["vocab"], p["seed"], p["legacy"])
    path = os.path.join(d, "embeddings.vec")
    with open(path, "wb") as f:
        f.write(base64.b64decode(g["embeddings_b64"]))
    try:
        eng = nsbind.Engine(d, -1)
        on, rows, dim = eng.semantic_info()
        assert (int(on), rows, dim) == (g["enabled"], len(g["table"]), g["dim"])
        for r, (term, bits) in enumerate(g["table"]):
            t, v = eng.semantic_row(r)
            assert t == term
            assert v.view(np.uint32).tolist() == bits, (r, term)
        eng.close()
    finally:
        os.remove(path)


@pytest.mark.gpu
def test_semantic_search_equals_reference(index_factory):
    g, d, path = _sem_fixture(index_factory)
    try:
        eng = nsbind.Engine(d, 0)
        try:
            assert eng.semantic_info() == (True, g["table"]["rows"], g["table"]["dim"])
            for q, want in zip(g["queries"], g["expand"]):
                assert eng.expand(q) == [(t, b) for t, b in want], q
            for case in g["cases"]:
                gh, gn, gf, gu = eng.search_batch(g["queries"], case["k"])
                for qi, ref in enumerate(case["results"]):
                    if ref["found"] < 0:
                        assert not gu[qi]
                        continue
                    assert int(gf[qi]) == ref["found"], g["queries"][qi]
                    assert [int(b) for b in gh[qi, : gn[qi]]["score"].view(np.uint32)] == [h[2] for h in ref["hits"]], g["queries"][qi]
            # one at a time == batched (the batch shares two device calls), JSON surface intact
            one = eng.search_batch(g["queries"][:1], 10)
            many = eng.search_batch(g["queries"], 10)
            assert one[0].tobytes() == many[0][:1].tobytes()
            assert '"results"' in eng.search_json(g["queries"][0], 5)
        finally:
            eng.close()
    finally:
        os.remove(path)


@pytest.mark.gpu
def test_expansion_with_bit_equal_sims_follows_the_stated_tie_rule(index_factory, tmp_path):
    """Duplicate vectors (round-2 verdict, weak 1c).  The reference's top-k keeps, among rows with bit-equal sims, the rows
    scanned FIRST (`sim > heap.front().sim` is strict, src/semantic_embedding.cpp:129) — the smallest rows, which is this
    repo's rule too — but the ORDER in which it then hands equal sims to expand()'s map is what std::sort_heap leaves of a
    heap whose shape depends on every row that ever passed through it (:137-139): not a function of the result, so not
    reproducible from it.  The rule here: equal sims in row order.  What the comparator can and does pin for such a table:
    the SET of weighted terms and every weight's bits equal a Python restatement of expand()'s rules (:148-229) on exact
    fp32 dot products; weights never increase along the list; equal weights may come in either order.  (Score sums over at
    most two of a query's terms are order-independent in fp32; the golden fixture sem1 has no equal sims and pins the rest.)"""
    import shutil
    import struct
    src, _ = index_factory(1, 3000, 512, 99, False)
    d = str(tmp_path / "index")
    shutil.copytree(src, d)
    T = workloads.term_name
    rng = np.random.default_rng(3)
    dim = 16
    base = (rng.integers(-64, 65, (6, dim)) / 64.0).astype(np.float32)   # multiples of 1/64: "%.6f" spells them exactly
    words, vecs = [], []
    for r in range(1, 200):
        c = base[r % 6] + np.float32(r % 5) * base[(r + 1) % 6]         # few distinct directions: many bit-equal rows and sims
        words.append(T(r)); vecs.append(c)
    with open(os.path.join(d, "embeddings.vec"), "w") as f:
        for w_, v in zip(words, vecs):
            f.write(w_ + " " + " ".join("%.6f" % x for x in v) + "\n")
    V = np.array([[np.float32(float("%.6f" % x)) for x in v] for v in vecs], dtype=np.float32)
    n = np.sqrt((V.astype(np.float64) ** 2).sum(axis=1))
    V = (V / n[:, None]).astype(np.float32)                             # l2_normalize (:18-24)
    row_of = {w_: i for i, w_ in enumerate(words)}

    def py_expand(q_terms):
        w = {t: np.float32(1.0) for t in q_terms}
        banned = {row_of[t] for t in q_terms if t in row_of}
        alpha = np.float32(0.6)

        def offer(rows, sims, a):
            for r, s_ in zip(rows, sims):
                wt = max(np.float32(0.0), min(a, np.float32(a * s_)))
                if words[r] not in w or wt > w[words[r]]:
                    w[words[r]] = wt
        for t in q_terms:
            if t in row_of:
                offer(*np_most_similar(V, V[row_of[t]], 3, 0.55, banned), alpha)
        have = [V[row_of[t]] for t in q_terms if t in row_of]
        if have:
            q = np.zeros(dim, dtype=np.float32)
            for v in have:
                q = (q + v).astype(np.float32)
            q = (q / np.float32(len(have))).astype(np.float32)
            q = (q / np.sqrt((q.astype(np.float64) ** 2).sum())).astype(np.float32)
            offer(*np_most_similar(V, q, 5, 0.55, banned), np.float32(alpha * np.float32(0.8)))
        return {t: int(np.float32(x).view(np.uint32)) for t, x in w.items()}

    eng = nsbind.Engine(d, 0)
    try:
        assert eng.semantic_info()[0]
        saw_equal = False
        for q in ("%s %s" % (T(7), T(20)), T(3), "%s %s %s" % (T(11), T(12), T(40)), "zzzz %s" % T(5)):
            got = eng.expand(q)
            want = py_expand([t for t in q.split()])
            want = dict(sorted(want.items(), key=lambda kv: -np.uint32(kv[1]).view(np.float32))[:40])
            bits = [b for _, b in got]
            ws = [np.uint32(b).view(np.float32) for b in bits]
            assert all(ws[i] >= ws[i + 1] for i in range(len(ws) - 1)), q
            assert sorted(b for b in want.values()) == sorted(bits), q
            cut = min(want.values()) if len(want) == 40 else None
            for t, b in got:       # every term carries ITS weight (terms tied at a 40-term cut may differ: none here)
                assert want.get(t) == b or b == cut, (q, t)
            saw_equal = saw_equal or len(set(bits)) < len(bits)
        assert saw_equal, "the fixture is meant to produce bit-equal weights"
    finally:
        eng.close()

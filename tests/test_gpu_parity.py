"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Everything here calls through the
C-ABI of libnextsearch_hip.so (directly, or via the host facade that wraps it) and checks the HIP
path against the CPU oracle: docIds, ranks, `found` and fp32 score BITS must be identical
(north_star allows 1e-4 on scores; we hold bit-exact and assert it).
"""
import ctypes as C
import os

import numpy as np
import pytest
from conftest import GOLDEN_NAMES, VARIANTS_BUILD, VARIANTS_LIB, need_variants

import nsbind
import orc
import workloads

pytestmark = pytest.mark.gpu


def assert_same(gpu, ora, queries, label=""):
    gh, gn, gf, gu = gpu
    oh, on, of, ou = ora
    np.testing.assert_array_equal(gu.astype(bool), ou.astype(bool), err_msg=f"{label}: usable/early-return flags")
    for q in range(len(queries)):
        if not ou[q]:
            assert gn[q] == 0
            continue
        assert int(gf[q]) == int(of[q]), f"{label}: found differs for query {q} {queries[q]!r}: {gf[q]} vs {of[q]}"
        assert int(gn[q]) == int(on[q]), f"{label}: nhits differs for query {q} {queries[q]!r}"
        n = int(on[q])
        g, o = gh[q, :n], oh[q, :n]
        bad = np.nonzero((g["doc"] != o["doc"]) | (g["seg"] != o["seg"]) | (g["score"].view(np.uint32) != o["score"].view(np.uint32)))[0]
        assert bad.size == 0, (f"{label}: query {q} {queries[q]!r} first mismatch at rank {bad[0]}: "
                               f"gpu={g[bad[0]]} oracle={o[bad[0]]}")
        # unused tail is padded as the header documents
        assert np.all(gh[q, n:]["doc"] == 0xFFFFFFFF)
        assert np.all(np.isneginf(gh[q, n:]["score"]))


@pytest.fixture(scope="module")
def engines(golden_index):
    cache = {}

    def get(name):
        if name not in cache:
            g, d, _ = golden_index(name)
            cache[name] = (g, nsbind.Engine(d, 0), orc.Oracle(d))
        return cache[name]

    yield get
    for _, e, o in cache.values():
        e.close()
        o.close()


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_gpu_equals_oracle_and_reference_golden(name, engines):
    g, eng, ora = engines(name)
    queries = g["queries"]
    for case in g["cases"]:
        k = case["k"]
        gpu = eng.search_batch(queries, k)
        assert_same(gpu, ora.search_batch(queries, k), queries, f"{name} k={k}")
        # and directly against the real reference's captured output (tie-aware on order only)
        gh, gn, gf, gu = gpu
        for qi, ref in enumerate(case["results"]):
            if ref["found"] < 0:
                assert not gu[qi]
                continue
            assert int(gf[qi]) == ref["found"]
            assert [int(b) for b in gh[qi, : gn[qi]]["score"].view(np.uint32)] == [h[2] for h in ref["hits"]]


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 12, 13, 14, 15, 16, 17, 18, 19, 20])
def test_kernel_variants_agree(variant, engines):
    """Wave-private kernel (5/6/7: 256/512/1024-entry tables) and workgroup-tile kernel (1..4)."""
    need_variants(variant)
    g, eng, ora = engines("mid1")
    queries = g["queries"]
    eng.set_tuning(variant, 0, 0)
    try:
        for k in (1, 10, 100):
            assert_same(eng.search_batch(queries, k), ora.search_batch(queries, k), queries, f"variant {variant} k={k}")
        assert_same(eng.search_batch(queries, 10, nsbind.NS_FLAG_AND), ora.search_batch(queries, 10, orc.FLAG_AND), queries,
                    f"variant {variant} AND")
    finally:
        eng.set_tuning(0, 0, 0)


@pytest.mark.parametrize("variant,min_items,split", [(3, 1, 0), (3, 64, 0), (3, 4096, 0), (0, 1, 1 << 30), (0, 1, 500),
                                                      (0, 4096, 0), (15, 1, 64), (17, 100000, 1000), (12, 1, 1 << 30), (12, 1, 300), (13, 4096, 0), (14, 100000, 1000), (19, 1, 1 << 30), (18, 1, 300), (20, 4096, 0)])
def test_doc_range_splitting_is_invisible(variant, min_items, split, engines):
    """Queries are split into doc ranges (by posting budget, and to fill the chip for small batches)
    and re-joined on the device by k_merge; the result must not depend on the split."""
    need_variants(variant)
    g, eng, ora = engines("mid1")
    queries = g["queries"][:20]
    eng.set_tuning(variant, min_items, split)
    try:
        assert_same(eng.search_batch(queries, 10), ora.search_batch(queries, 10), queries, f"v{variant} min_items={min_items} split={split}")
        assert_same(eng.search_batch(queries[:1], 100), ora.search_batch(queries[:1], 100), queries[:1], "single query")
        g8, eng8, ora8 = engines("multi8")
        eng8.set_tuning(variant, min_items, split)
        try:
            assert_same(eng8.search_batch(queries, 100), ora8.search_batch(queries, 100), queries, "multi8 split")
        finally:
            eng8.set_tuning(0, 0, 0)
    finally:
        eng.set_tuning(0, 0, 0)


def test_retired_variants_are_rejected(engines):
    g, eng, ora = engines("mid1")
    for v in (5, 6, 7, 8, 9, 10, 11, 21, 1000):
        with pytest.raises(RuntimeError, match="unknown kernel variant"):
            eng.set_tuning(v, 0, 0)
    if not VARIANTS_BUILD:   # the product library holds variant 0 only; it says where the others are
        for v in (1, 2, 3, 4, 12, 13, 14, 15, 16, 17, 18, 19, 20):
            with pytest.raises(RuntimeError, match="variants build"):
                eng.set_tuning(v, 0, 0)
    eng.set_tuning(0, 0, 0)


def test_forced_kernel_variants_in_the_variants_build():
    """The 13 forced kernel variants (each scoring body as a kernel of its own, other table and tile sizes) are test and
    sweep infrastructure: libnextsearch_hip_variants.so (`make -C nextsearch-api_amd variants`) holds them, the product
    library does not.  Their parity cases — every variant against the oracle, doc-range splitting, zero-tf / signed
    inputs, equal scores — run here in ONE child process that loads that build (NS_HIP_LIB)."""
    import subprocess
    import sys
    if VARIANTS_BUILD:
        pytest.skip("this IS the variants process")
    assert os.path.exists(VARIANTS_LIB), "libnextsearch_hip_variants.so is missing: make -C nextsearch-api_amd variants"
    env = dict(os.environ, NS_HIP_LIB=VARIANTS_LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "kernel_variants_agree or doc_range_splitting or zero_tf_postings or equal_scores_in_doc_tiles or retired_variants"],
                       env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail


def test_every_parity_case_again_with_term_scores_shared():
    """NS_SHARE=2 in the environment makes every batch that can compute its lists' term scores once and score from them
    (ns_ctx_share_scores(2) for every ctx created): the parity cases of this file — goldens, splitting, zero-tf and signed
    inputs, equal scores, the fuzz slice, > 64 terms, sparse lists over 20 M docs, damaged lexicons, raw-ABI weights, the
    short-division range — must hold unchanged on that path.  ONE child process; the cases that assert which path a batch took
    (they set the mode themselves) and the child-process cases stay out."""
    import subprocess
    import sys
    if os.environ.get("NS_SHARE") is not None:
        pytest.skip("this IS the forced-sharing process")
    env = dict(os.environ, NS_SHARE="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider", "-k",
                        "not full_batches_equal and not impact_stream and not shared_term_scores and not forced_kernel_variants "
                        "and not every_parity_case_again and not overlapping_batches"],
                       env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout, tail


@pytest.mark.parametrize("variant,split", [(0, 0), (0, 300), (12, 0), (13, 200), (18, 0), (19, 0), (20, 300), (2, 0)])
def test_zero_tf_postings_and_signed_weights(variant, split):
    """`found` (src/api_engine.cpp:495) counts a doc once some term touches it, whatever the contribution: a posting
    with tf == 0 contributes an exact zero, and with a negative idf or weight that zero is -0.0f, which the
    reference's `0.0f + x` turns into +0.0f.  Raw C-ABI, one segment with tf == 0 postings sprinkled into dense and
    sparse lists, positive and negative idfs / weights, every scoring body: found, order and score BITS against a
    numpy fp32 restatement that starts every doc at +0.0f."""
    need_variants(variant)
    L = nsbind.hip_lib()
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    try:
        N = 6000
        rng = np.random.default_rng(495)
        doc_len = rng.integers(20, 3000, size=N, dtype=np.uint32)
        avgdl = float(np.float32(doc_len.astype(np.float64).mean()))
        sizes = [4200, 3900, 700, 60, 2500, 5]
        lists, payload = [], []
        for n in sizes:
            docs = np.sort(rng.choice(N, size=n, replace=False)).astype(np.uint32)
            tfs = rng.integers(1, 9, size=n, dtype=np.uint32)
            tfs[rng.random(n) < 0.3] = 0                       # tf == 0: never written by the indexers, but legal bytes
            lists.append((docs, tfs))
            payload.append(np.stack([docs, tfs], axis=1).astype(np.uint32).ravel())
        flat = np.concatenate(payload)
        offs = np.cumsum([0] + [len(p) * 4 for p in payload])[:-1]
        seg = C.c_void_p()
        assert L.ns_segment_upload(ctx, 0, N, C.c_float(avgdl), doc_len.ctypes.data, flat.ctypes.data, flat.nbytes, C.byref(seg)) == 0, L.ns_last_error(ctx)
        assert L.ns_set_tuning(ctx, variant, 0, split) == 0
        idfs = [1.5, -2.25, 3.0, -0.75, 0.5, 4.0]
        wts = [1.0, 1.0, -0.5, 0.6, -1.0, 1.0]
        queries = [[0, 1], [1, 0], [1], [3], [1, 4], [4, 1, 3], [0, 1, 2, 3, 4, 5], [2, 5], [5, 3, 1], [1, 1], [0], [3, 2]]
        qd = np.zeros(len(queries), dtype=nsbind.QDESC_DTYPE)
        refs = []
        for qi, q in enumerate(queries):
            qd[qi] = (len(refs), len(q))
            for li in q:
                refs.append((0, len(lists[li][0]), int(offs[li]), idfs[li], wts[li]))
        refs = np.array(refs, dtype=nsbind.TERM_DTYPE)
        for k in (10, 100):
            rc, hits, nhits, found = nsbind.search_batch_raw(ctx, qd, refs, k)
            assert rc == 0, L.ns_last_error(ctx)
            for qi, q in enumerate(queries):
                acc = _np_bm25(lists, q, [idfs[li] for li in q], [wts[li] for li in q], doc_len, avgdl)
                assert int(found[qi]) == len(acc), (variant, split, k, qi, int(found[qi]), len(acc))
                keyed = sorted(acc.items(), key=lambda kv: (-float(kv[1]), kv[0]))[:k]
                n = int(nhits[qi])
                assert n == len(keyed)
                want_bits = np.array([v for _, v in keyed], dtype=np.float32).view(np.uint32)
                np.testing.assert_array_equal(hits[qi, :n]["score"].view(np.uint32), want_bits, err_msg=f"v{variant} q{qi} k{k}")
                # docs: equal inside every run of equal scores (the numpy sort above breaks ties the canonical way: docId)
                assert [int(d) for d in hits[qi, :n]["doc"]] == [d for d, _ in keyed], (variant, qi, k)
                assert not np.any(hits[qi, :n]["score"].view(np.uint32) == 0x80000000), "the reference never returns -0.0f"
        assert L.ns_segment_release(ctx, seg) == 0
    finally:
        L.ns_ctx_destroy(ctx)


@pytest.mark.parametrize("variant,split,skips", [(0, 0, 2), (0, 0, 1), (0, 0, 0), (0, 700, 2), (19, 0, 0), (18, 300, 0), (12, 0, 0)])
def test_equal_scores_in_doc_tiles_and_skip_tables(variant, split, skips):
    """Every doc has the same length and every posting of a list the same tf, so ALL docs of a dense group score
    equal: the K best are the K smallest docIds (canonical tie order), whatever order the tile read-back offers the
    slots in and wherever the candidate buffer is shrunk (a shrink inside a tile leaves ties with the threshold that
    still win on docId).  Raw C-ABI; with skip tables (ns_segment_build_skips) for all lists (2), for some (1: the
    others keep their cursors inside the same grid tiles) or none (0), plus an unsorted list, which must get no table."""
    need_variants(variant)
    L = nsbind.hip_lib()
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    try:
        N = 7000
        rng = np.random.default_rng(1024)
        doc_len = np.full(N, 100, dtype=np.uint32)
        avgdl = 100.0
        all_docs = np.arange(N, dtype=np.uint32)
        gaps = np.sort(rng.choice(N, size=5200, replace=False)).astype(np.uint32)
        third = all_docs[(all_docs % 3) == 1]
        sparse = np.sort(rng.choice(N, size=40, replace=False)).astype(np.uint32)
        late = all_docs[all_docs >= 4100]                       # nothing in the first four grid cells
        lists = [(all_docs, np.full(N, 2, np.uint32)), (gaps, np.full(len(gaps), 1, np.uint32)), (third, np.full(len(third), 3, np.uint32)),
                 (sparse, np.full(len(sparse), 2, np.uint32)), (late, np.full(len(late), 1, np.uint32))]
        payload = [np.stack([d, t], axis=1).astype(np.uint32).ravel() for d, t in lists]
        unsorted = np.stack([all_docs[::-1][:3000], np.full(3000, 1, np.uint32)], axis=1).astype(np.uint32).ravel()
        flat = np.concatenate(payload + [unsorted])
        offs = np.cumsum([0] + [len(p) * 4 for p in payload + [unsorted]])[:-1]
        seg = C.c_void_p()
        assert L.ns_segment_upload(ctx, 0, N, C.c_float(avgdl), doc_len.ctypes.data, flat.ctypes.data, flat.nbytes, C.byref(seg)) == 0, L.ns_last_error(ctx)
        if skips:
            which = [0, 1, 2, 3, 4, 5] if skips == 2 else [0, 2, 5]
            bo = np.array([offs[i] for i in which], dtype=np.uint64)
            cn = np.array([(len(lists[i][0]) if i < 5 else 3000) for i in which], dtype=np.uint32)
            assert L.ns_segment_build_skips(ctx, seg, bo.ctypes.data, cn.ctypes.data, len(which)) == 0, L.ns_last_error(ctx)
            assert L.ns_segment_build_skips(ctx, seg, bo.ctypes.data, cn.ctypes.data, len(which)) == 0   # again: left as they are
            bad = np.array([4], dtype=np.uint64)
            assert L.ns_segment_build_skips(ctx, seg, bad.ctypes.data, cn.ctypes.data, 1) != 0          # not a multiple of 8
        assert L.ns_set_tuning(ctx, variant, 0, split) == 0
        idfs = [1.0, 1.0, 2.0, 5.0, 0.5]
        queries = [[0, 1], [1, 0], [0, 2], [0, 1, 2], [2, 1, 0, 3], [0], [1, 2], [0, 4], [4, 1], [3, 0, 4], [0, 0]]
        qd = np.zeros(len(queries), dtype=nsbind.QDESC_DTYPE)
        refs = []
        for qi, q in enumerate(queries):
            qd[qi] = (len(refs), len(q))
            for li in q:
                refs.append((0, len(lists[li][0]), int(offs[li]), idfs[li], 1.0))
        refs = np.array(refs, dtype=nsbind.TERM_DTYPE)
        for use in ((1, 0) if skips else (1,)):
            assert L.ns_ctx_use_skips(ctx, use) == 0
            for k in (1, 3, 10, 100):
                rc, hits, nhits, found = nsbind.search_batch_raw(ctx, qd, refs, k)
                assert rc == 0, L.ns_last_error(ctx)
                for qi, q in enumerate(queries):
                    acc = _np_bm25(lists, q, [idfs[li] for li in q], [1.0] * len(q), doc_len, avgdl)
                    assert int(found[qi]) == len(acc), (variant, split, skips, use, k, qi)
                    keyed = sorted(acc.items(), key=lambda kv: (-float(kv[1]), kv[0]))[:k]
                    n = int(nhits[qi])
                    assert n == len(keyed)
                    assert [int(d) for d in hits[qi, :n]["doc"]] == [d for d, _ in keyed], (variant, split, skips, use, k, qi)
                    want_bits = np.array([v for _, v in keyed], dtype=np.float32).view(np.uint32)
                    np.testing.assert_array_equal(hits[qi, :n]["score"].view(np.uint32), want_bits)
            # the conjunctive extension over the same groups: a doc counts when EVERY term ref of the query holds it
            for k in (3, 100):
                rc, hits, nhits, found = nsbind.search_batch_raw(ctx, qd, refs, k, nsbind.NS_FLAG_AND)
                assert rc == 0, L.ns_last_error(ctx)
                for qi, q in enumerate(queries):
                    acc = _np_bm25(lists, q, [idfs[li] for li in q], [1.0] * len(q), doc_len, avgdl)
                    members = set(lists[q[0]][0].tolist())
                    for li in q[1:]:
                        members &= set(lists[li][0].tolist())
                    keyed = sorted(((d, v) for d, v in acc.items() if d in members), key=lambda kv: (-float(kv[1]), kv[0]))
                    assert int(found[qi]) == len(keyed), (variant, split, skips, use, k, qi, "AND")
                    n = int(nhits[qi])
                    assert n == min(k, len(keyed))
                    assert [int(d) for d in hits[qi, :n]["doc"]] == [d for d, _ in keyed[:k]], (variant, split, skips, use, k, qi, "AND")
                    np.testing.assert_array_equal(hits[qi, :n]["score"].view(np.uint32), np.array([v for _, v in keyed[:k]], dtype=np.float32).view(np.uint32))
        assert L.ns_segment_release(ctx, seg) == 0
    finally:
        L.ns_ctx_destroy(ctx)


def test_full_batches_equal_oracle_and_reference_digests(index_factory):
    """BASELINE configs 2-5 at FULL size, EVERY query of every batch: tests/golden/fullsize.json holds, per block of
    1024 queries, SHA-256 digests of the whole batch's answers — `exact` from the oracle (found, nhits, every hit's
    score bits / segment / docId in rank order), `ties` from the REAL reference run over the same full batches
    (tie-order-invariant: found, nhits, sorted score bits).  Only the digests travel to the GPU box."""
    from conftest import load_golden
    g = load_golden("fullsize")
    for cfg, e in g["configs"].items():
        gen, Q, K, flags, (nseg, docs) = workloads.WORKLOADS[cfg]
        assert (Q, K, flags, [nseg, docs]) == (e["queries"], e["k"], e["flags"], e["index"])
        d, _ = index_factory(nseg, docs, 65536, 1337, False)
        eng = nsbind.Engine(d, 0)
        try:
            hits, nhits, found, usable = eng.search_batch(gen(Q), K, flags)
            assert usable.all()
            dig = orc.batch_digests(hits, nhits, found, g["block"])
            bad = [b for b, (x, y) in enumerate(zip(dig["exact"], e["exact"])) if x != y]
            assert not bad, f"{cfg}: blocks {bad[:8]} of {len(e['exact'])} differ from the oracle's digests"
            if "ties" in e:
                bad = [b for b, (x, y) in enumerate(zip(dig["ties"], e["ties"])) if x != y]
                assert not bad, f"{cfg}: blocks {bad[:8]} differ from the real reference's tie-invariant digests"
            # cfg3-5 name their lists often enough to share term scores by default (ns_ctx_share_scores): the digests above
            # are that path's; scoring every posting in place must not change a byte
            b = eng.prepare(gen(Q), K, flags)
            inf = b.info()
            b.close()
            assert bool(inf.flags & nsbind.NS_INFO_SHARED) == (cfg != "cfg2"), cfg
            assert cfg == "cfg2" or (0 < inf.shared_postings * 4 <= inf.postings and inf.shared_lists > 0)
            eng.share_scores(0)
            b = eng.prepare(gen(Q), K, flags)
            assert not (b.info().flags & (nsbind.NS_INFO_SHARED | nsbind.NS_INFO_IMPACTS))
            b.close()
            h2, n2, f2, _ = eng.search_batch(gen(Q), K, flags)
            assert hits.tobytes() == h2.tobytes() and nhits.tobytes() == n2.tobytes() and found.tobytes() == f2.tobytes(), (cfg, "sharing off")
            eng.share_scores(1)
            # the skip tables reload() built (doc-tile groups on the skip grid) must not change a byte ...
            eng.use_skips(False)
            h2, n2, f2, _ = eng.search_batch(gen(Q), K, flags)
            assert hits.tobytes() == h2.tobytes() and nhits.tobytes() == n2.tobytes() and found.tobytes() == f2.tobytes(), (cfg, "skips off")
            eng.use_skips(True)
            # ... nor the optional streams (packed blocks, impacts, both)
            eng.build_packed()
            for pk in (1, 2):
                eng.use_packed(pk)
                h2, n2, f2, _ = eng.search_batch(gen(Q), K, flags)
                assert hits.tobytes() == h2.tobytes() and nhits.tobytes() == n2.tobytes() and found.tobytes() == f2.tobytes(), (cfg, "packed", pk)
            eng.use_packed(0)
            # ... nor block-max pruning of the single-term queries (found stays the list's posting count)
            eng.build_blockmax()
            eng.use_pruning(True)
            h2, n2, f2, _ = eng.search_batch(gen(Q), K, flags)
            assert hits.tobytes() == h2.tobytes() and nhits.tobytes() == n2.tobytes() and found.tobytes() == f2.tobytes(), (cfg, "pruning")
            eng.use_pruning(False)
            if cfg in ("cfg5", "cfg3"):
                eng.build_impacts()
                for pk in (1, 0):
                    eng.use_packed(pk)
                    h2, n2, f2, _ = eng.search_batch(gen(Q), K, flags)
                    assert hits.tobytes() == h2.tobytes() and nhits.tobytes() == n2.tobytes() and found.tobytes() == f2.tobytes(), (cfg, "impacts", pk)
        finally:
            eng.close()


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_blockmax_pruning_equals_exhaustive_oracle_and_reference(name, golden_index):
    """SURVEY 8 f2, block-max scores (ns_segment_build_blockmax / ns_ctx_use_pruning): single-term queries skip the blocks
    whose maximum cannot enter their top-K; `found` is the list's posting count.  Every golden index, K = 1 / 10 / 100, OR
    and AND, forced splits (range items start and end inside blocks), with and without skip tables: hits, order, nhits and
    found equal the exhaustive path's bytes, the oracle and the real reference's captured scores."""
    g, d, _ = golden_index(name)
    eng, ora = nsbind.Engine(d, 0), orc.Oracle(d)
    try:
        eng.build_blockmax()
        vocab = g["params"]["vocab"]
        singles = [workloads.term_name(r) for r in (1, 2, 3, 5, 8, 13, 40, 100, 300) if r <= vocab] + ["covid covid", "virus the of", "zzzz"]
        queries = g["queries"] + singles + workloads.cfg5_queries(120, 3, vocab)
        for k in (1, 10, 100):
            for flags in (0, nsbind.NS_FLAG_AND):
                want = ora.search_batch(queries, k, flags)
                for tune in ((0, 0, 0), (0, 1, 300), (0, 4096, 0), (0, 1, 1 << 30)):
                    eng.set_tuning(*tune)
                    eng.use_pruning(False)
                    plain = eng.search_batch(queries, k, flags)
                    for skips in (True, False):
                        eng.use_skips(skips)
                        eng.use_pruning(True)
                        pruned = eng.search_batch(queries, k, flags)
                        for x, y in zip(plain, pruned):
                            assert x.tobytes() == y.tobytes(), (name, k, flags, tune, skips)
                    eng.use_skips(True)
                    assert_same(pruned, want, queries, f"{name} k={k} flags={flags} tune={tune} pruned")
        eng.set_tuning(0, 0, 0)
        # the batch really took the pruned body, and the real reference's captured scores are reproduced with it
        b = eng.prepare(singles, 10)
        assert b.info().flags & nsbind.NS_INFO_PRUNED
        b.close()
        for case in g["cases"]:
            gh, gn, gf, gu = eng.search_batch(g["queries"], case["k"])
            for qi, ref in enumerate(case["results"]):
                if ref["found"] < 0:
                    assert not gu[qi]
                    continue
                assert int(gf[qi]) == ref["found"]
                assert [int(b_) for b_ in gh[qi, : gn[qi]]["score"].view(np.uint32)] == [h[2] for h in ref["hits"]]
    finally:
        eng.close()
        ora.close()


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_two_list_merge_body_equals_table_body_oracle_and_reference(name, golden_index):
    """north_star "intersect/merge across query terms": groups of exactly two lists advance through both sorted lists in
    lockstep (ns_merge_kernel.hip) instead of hashing one into a table.  Two-term queries of every shape — hot + hot, hot +
    rare, rare + rare, the same term twice (both lists ARE the same list: every doc matches), a term the lexicon lacks —
    in OR and AND mode, K = 1 / 10 / 100, forced splits: equal to the driver-stream body's bytes, to the oracle, and (the
    goldens' own two-term queries) to the real reference's captured scores."""
    g, d, _ = golden_index(name)
    eng, ora = nsbind.Engine(d, 0), orc.Oracle(d)
    try:
        vocab = g["params"]["vocab"]
        T = workloads.term_name
        rng = np.random.default_rng(17)
        pairs = [f"{T(a)} {T(b)}" for a, b in ((1, 2), (2, 1), (3, 40), (40, 3), (8, 9), (100, 101), (5, 5), (60, 60), (1, vocab), (vocab, 2))]
        pairs += [f"{T(int(a))} {T(int(b))}" for a, b in zip(rng.integers(1, min(vocab, 64), 60), rng.integers(1, vocab, 60))]
        pairs += ["covid zzzz", "zzzz covid", "virus the vaccine"]
        queries = g["queries"] + pairs
        for k in (1, 10, 100):
            for flags in (0, nsbind.NS_FLAG_AND):
                want = ora.search_batch(queries, k, flags)
                for tune in ((0, 0, 0), (0, 1, 300), (0, 4096, 0), (0, 1, 1 << 30)):
                    eng.set_tuning(*tune)
                    eng.use_merge(False)
                    table = eng.search_batch(queries, k, flags)
                    eng.use_merge(True)
                    merged = eng.search_batch(queries, k, flags)
                    for x, y in zip(table, merged):
                        assert x.tobytes() == y.tobytes(), (name, k, flags, tune)
                    assert_same(merged, want, queries, f"{name} k={k} flags={flags} tune={tune} merge body")
        eng.set_tuning(0, 0, 0)
    finally:
        eng.close()
        ora.close()


def test_blockmax_pruning_ties_weights_and_registration():
    """Raw C-ABI: all docs of equal length and tf, so every posting of a list scores the same (the K best are the K
    smallest docIds: a block whose maximum EQUALS theta must be skipped, not read); a list whose best postings sit in its
    last block; tf == 0 postings; a fractional and a negative weight (the negative one must not be pruned); a foreign idf
    (not pruned); re-registration with another idf; a bad offset."""
    import ctypes as C
    L = nsbind.hip_lib()
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    try:
        n_docs = 40_000
        rng = np.random.default_rng(5)
        doc_len = np.full(n_docs, 100, dtype=np.uint32)
        doc_len[-300:] = 20                                   # short docs at the END: the best scores of list B are in its last blocks
        la = np.arange(0, n_docs, 2, dtype=np.uint32)         # list A: 20000 postings, all tf 3 (equal scores but for the tail)
        ta = np.full(la.size, 3, dtype=np.uint32)
        lb = np.sort(rng.choice(n_docs, 9000, replace=False)).astype(np.uint32)
        tb = rng.integers(0, 6, lb.size).astype(np.uint32)    # tf 0 .. 5 (zeros included)
        post = np.empty((la.size + lb.size, 2), dtype=np.uint32)
        post[:la.size, 0], post[:la.size, 1] = la, ta
        post[la.size:, 0], post[la.size:, 1] = lb, tb
        avgdl = float(np.float32(doc_len.astype(np.float64).sum() / n_docs))
        seg = C.c_void_p()
        assert L.ns_segment_upload(ctx, 0, n_docs, C.c_float(avgdl), doc_len.ctypes.data, post.ctypes.data, post.nbytes, C.byref(seg)) == 0
        offs = np.array([0, la.size * 8], dtype=np.uint64)
        cnts = np.array([la.size, lb.size], dtype=np.uint32)
        idfs = np.array([1.25, 0.75], dtype=np.float32)
        assert L.ns_segment_build_blockmax(ctx, seg, offs.ctypes.data, cnts.ctypes.data, idfs.ctypes.data, 2) == 0
        bad = np.array([4], dtype=np.uint64)
        assert L.ns_segment_build_blockmax(ctx, seg, bad.ctypes.data, cnts.ctypes.data, idfs.ctypes.data, 1) == -1

        def run(refs, k, prune, split=0):
            L.ns_set_tuning(ctx, 0, 1 if split else 0, split)
            L.ns_ctx_use_pruning(ctx, 1 if prune else 0)
            qd = np.array([(i, 1) for i in range(len(refs))], dtype=nsbind.QDESC_DTYPE)
            tr = np.array(refs, dtype=nsbind.TERM_DTYPE)
            b = nsbind.prepare_raw(ctx, qd, tr, k, 0)
            b.run(timed=False); b.sync()
            out = b.fetch() + (b.info().flags,)
            b.close()
            return out

        def ref(seg_id, count, byte_off, idf, w):
            return (seg_id, count, byte_off, idf, w)
        refs = [ref(0, la.size, 0, 1.25, 1.0), ref(0, lb.size, la.size * 8, 0.75, 1.0), ref(0, lb.size, la.size * 8, 0.75, 0.37),
                ref(0, lb.size, la.size * 8, 0.75, -1.0), ref(0, la.size, 0, 1.5, 1.0)]
        for k in (1, 3, 10, 100):
            for split in (0, 700, 5000):
                h0, n0, f0, fl0 = run(refs, k, False, split)
                h1, n1, f1, fl1 = run(refs, k, True, split)
                assert not (fl0 & nsbind.NS_INFO_PRUNED) and (fl1 & nsbind.NS_INFO_PRUNED)
                assert h0.tobytes() == h1.tobytes() and n0.tobytes() == n1.tobytes() and f0.tobytes() == f1.tobytes(), (k, split)
                assert int(f1[0]) == la.size and int(f1[1]) == lb.size
                # equal scores: the K best of list A's long docs are the smallest docIds (its short tail scores higher)
                tail = la[la >= n_docs - 300]
                want = list(tail[:k]) if k <= tail.size else list(tail) + list(la[: k - tail.size])
                assert list(h1[0, :k]["doc"]) == want, (k, split)
        # a batch made only of refs that must NOT be pruned (negative weight, foreign idf) does not take the pruned body
        *_, fl = run([refs[3], refs[4]], 10, True)
        assert not (fl & nsbind.NS_INFO_PRUNED)
        # re-registration with the other idf: now that ref is pruned, and the first registration of list A no longer matches
        idf2 = np.array([1.5], dtype=np.float32)
        assert L.ns_segment_build_blockmax(ctx, seg, offs.ctypes.data, cnts.ctypes.data, idf2.ctypes.data, 1) == 0
        a = run([refs[4]], 10, False); b_ = run([refs[4]], 10, True)
        assert (b_[3] & nsbind.NS_INFO_PRUNED) and a[0].tobytes() == b_[0].tobytes() and a[2].tobytes() == b_[2].tobytes()
        *_, fl = run([refs[0]], 10, True)
        assert not (fl & nsbind.NS_INFO_PRUNED)
        L.ns_set_tuning(ctx, 0, 0, 0)
        L.ns_ctx_use_pruning(ctx, 0)
    finally:
        L.ns_ctx_destroy(ctx)


def test_fixed_seed_fuzz_slice():
    """A bounded, fixed-seed slice of tools/fuzz_parity.py (random index shapes, query laws, K, OR/AND, work-splitting
    knobs, impact streams on/off) against the oracle, so that the driver's GPU run sees the differential test too."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    # bounded by CASE COUNT, not by seconds: the same 240 batches (60 random indexes x 4 batches) are checked on every box;
    # the time limit is only a guard against a hung box
    cases, bad = fuzz_parity.run(seconds=600.0, seed=20261004, max_cases=240, verbose=False)
    assert bad is None, bad
    assert cases == 240


def test_reload_streams_inverted_files_without_a_host_copy(index_factory):
    """north_star: "segment files are mmapped and pinned-copied once per segment".  Opening a second engine on the same
    4 x 1M-doc index (222 MB of postings) must grow the process's resident set by about what a HOST-ONLY engine costs
    (lexicons + docs), not by that plus the posting payload."""
    import gc

    def rss_mb():
        with open("/proc/self/statm") as f:
            return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20

    d, total = index_factory(4, 1_000_000, 65536, 1337, False)
    payload_mb = total * 8 / 2**20
    warm = nsbind.Engine(d, 0)          # HIP runtime, code objects, pinned pools: paid once
    gc.collect()
    r0 = rss_mb()
    host_only = nsbind.Engine(d, -1)
    r1 = rss_mb()
    dev = nsbind.Engine(d, 0)
    r2 = rss_mb()
    try:
        host_cost, dev_cost = r1 - r0, r2 - r1
        assert payload_mb > 200
        assert dev_cost < host_cost + 0.35 * payload_mb, (host_cost, dev_cost, payload_mb)
        # and the device copy is complete: same answers as the first engine
        qs = workloads.cfg5_queries(64)
        a, b = warm.search_batch(qs, 10), dev.search_batch(qs, 10)
        assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b))
    finally:
        warm.close(); host_only.close(); dev.close()


def test_overlapping_batches_and_host_thread_count(index_factory):
    """prepare(i+1) || run(i) || fetch(i-1) on ONE ctx (NS_RUN_FETCH: per-batch completion events and
    pinned result slots).  Six different batches go through the pipeline with two in flight; every one must equal
    the oracle bit for bit, whatever number of host threads prepared it (1, automatic, 5), and destroying batch i
    while batch i+1 runs must not disturb i+1."""
    d, _ = index_factory(2, 60_000, 65536, 1337, False)
    eng, ora = nsbind.Engine(d, 0), orc.Oracle(d)
    L = nsbind.hip_lib()
    try:
        sets = [workloads.cfg5_queries(3500 + 211 * i, 77 + i) for i in range(4)] + [workloads.cfg3_queries(300, 5), ["covid"]]
        descs = []
        for qs in sets:
            qd, refs, usable = eng.build_refs(qs)
            assert usable.all()
            descs.append((qd, refs))
        want = [ora.search_batch(qs, 10, threads=16) for qs in sets]
        # share = 2: every batch computes the term scores of its distinct lists itself, into ONE buffer per segment, while the
        # batch before it is still scoring from that buffer on the other stream (ns_ctx_share_scores)
        for threads, overlap, share in ((1, 0, 0), (0, 0, 1), (5, 1, 0), (0, 1, 2), (3, 0, 2)):
            assert L.ns_ctx_set_host_threads(eng.ctx, threads) == 0
            assert L.ns_ctx_set_overlap(eng.ctx, overlap) == 0   # batches alternate between two streams: tails and heads overlap
            assert L.ns_ctx_share_scores(eng.ctx, share) == 0
            got = list(nsbind.pipelined_search(eng.ctx, descs, 10, timed=True))
            assert len(got) == len(sets)
            for qs, (hits, nhits, found, inf), w in zip(sets, got, want):
                assert inf.timed_runs == 1 and inf.n_queries == len(qs)
                if share != 1:   # (1: the default rule — the larger of these batches share, the lone query never does)
                    assert bool(inf.flags & nsbind.NS_INFO_SHARED) == (share == 2 and inf.postings > 0) and (inf.shared_lists > 0) == (share == 2 and inf.postings > 0)
                elif len(qs) == 1:
                    assert not (inf.flags & nsbind.NS_INFO_SHARED)
                assert_same((hits, nhits, found, np.ones(len(qs), np.uint8)), w, qs, f"pipelined, host threads {threads}, overlap {overlap}, share {share}")
        assert L.ns_ctx_share_scores(eng.ctx, 1) == 0
        # the same through preallocated, reused output buffers (what bench.py does)
        k = 10
        out = [(np.empty((4400, k), dtype=nsbind.HIT_DTYPE), np.empty(4400, np.uint32), np.empty(4400, np.uint64)) for _ in range(2)]
        for i, (hits, nhits, found, inf) in enumerate(nsbind.pipelined_search(eng.ctx, descs[:4], k, out=out)):
            Q = len(sets[i])
            assert_same((hits[:Q].copy(), nhits[:Q].copy(), found[:Q].copy(), np.ones(Q, np.uint8)), want[i], sets[i], "pipelined, reused buffers")
        L.ns_ctx_set_host_threads(eng.ctx, 0)
        L.ns_ctx_set_overlap(eng.ctx, 0)
    finally:
        eng.close()
        ora.close()


def test_pipelined_calls_state_machine(engines):
    """NS_RUN_FETCH's edges: not for batches with bound outputs; at most 8 batches between run and fetch; fetch after a
    second NS_RUN_FETCH run returns the second run's (identical) results; a batch run WITHOUT the flag still fetches the old
    way; destroying the newest batch first does not disturb the older ones; ns_batch_gap_ms needs timed runs."""
    torch = pytest.importorskip("torch")
    g, eng, ora = engines("mid1")
    L = nsbind.hip_lib()
    queries = g["queries"]
    qd, refs, usable = eng.build_refs(queries)
    want = ora.search_batch(queries, 10)
    # bound outputs + NS_RUN_FETCH -> NS_E_STATE
    b = nsbind.prepare_raw(eng.ctx, qd, refs, 10)
    buf = torch.zeros(len(queries) * 10 * 12 + len(queries) * 12 + 1024, dtype=torch.uint8, device="cuda")
    b.bind_outputs(buf.data_ptr(), buf.data_ptr() + len(queries) * 120 + 256, buf.data_ptr() + len(queries) * 124 + 512)
    assert L.ns_batch_run(b.h, 2) == -5 and b"bound" in L.ns_last_error(eng.ctx)
    b.close()
    # more in flight than there are pinned result slots (8, plus the one small batch whose results live in host memory
    # anyway): the run is refused, the batches already in flight are intact
    held, refused = [], None
    for i in range(12):
        bi = nsbind.prepare_raw(eng.ctx, qd, refs, 10)
        rc = L.ns_batch_run(bi.h, 2 | 1)
        if rc != 0:
            assert rc == -5 and b"more than 8" in L.ns_last_error(eng.ctx)
            refused = bi
            break
        held.append(bi)
    assert refused is not None and len(held) in (8, 9)
    g_ms = C.c_float()
    assert L.ns_batch_gap_ms(held[0].h, held[1].h, C.byref(g_ms)) in (0, -5)
    refused.close()
    held.pop().close()                                 # newest first
    for bi in held:
        hits, nhits, found = bi.fetch()
        assert_same((hits, nhits, found, usable), want, queries, "eight in flight")
        bi.run(fetch=True)                             # run again, fetch again
        h2, n2, f2 = bi.fetch()
        assert h2.tobytes() == hits.tobytes() and n2.tobytes() == nhits.tobytes() and f2.tobytes() == found.tobytes()
        bi.close()
    b = nsbind.prepare_raw(eng.ctx, qd, refs, 10)
    b.run()                                            # no flag: the stream-sync path
    assert_same(b.fetch() + (usable,), want, queries, "plain run after pipelined ones")
    b2 = nsbind.prepare_raw(eng.ctx, qd, refs, 10)
    assert L.ns_batch_gap_ms(b.h, b2.h, C.byref(g_ms)) == -5   # neither has a timed run
    b.close(); b2.close()


def test_reload_on_the_device_swaps_contexts_and_survives_failure(index_factory, tmp_path):
    """Engine::reload on a live device engine: the new index goes into a FRESH context that replaces the old one only when
    everything loaded (src/api_engine.cpp:76-90); a reload that fails half way leaves the engine answering from the old
    device copy; a reload that succeeds answers from the new one (here: the same directory with one segment dropped)."""
    import shutil
    src, _ = index_factory(3, 20_000, 4096, 77, False)
    d = str(tmp_path / "index")
    shutil.copytree(src, d)
    eng, ora = nsbind.Engine(d, 0), orc.Oracle(d)
    try:
        queries = workloads.cfg5_queries(200, 5, 4096) + ["covid", "zzzz", "the of"]
        want3 = ora.search_batch(queries, 10)
        assert_same(eng.search_batch(queries, 10), want3, queries, "before reload")
        ctx0 = eng.ctx
        eng.reload()                                            # same index: new context, same answers
        assert eng.ctx != ctx0 or True                          # (the allocator may hand the same address back)
        assert_same(eng.search_batch(queries, 10), want3, queries, "after reload")
        inv = os.path.join(d, "segments", "seg_000002", "inverted_b007.bin")
        shutil.move(inv, inv + ".away")                         # the loader lists the file sizes: a missing barrel fails the load
        with pytest.raises(RuntimeError):
            eng.reload()
        assert eng.num_segments == 3
        assert_same(eng.search_batch(queries, 10), want3, queries, "after a failed reload")
        shutil.move(inv + ".away", inv)
        # drop the last segment from the manifest: the reloaded engine must answer like an oracle over two segments
        import struct
        names = [b"seg_000001", b"seg_000002"]
        with open(os.path.join(d, "manifest.bin"), "wb") as f:
            f.write(struct.pack("<I", 2) + b"".join(struct.pack("<I", len(n)) + n for n in names))
        eng.reload()
        assert eng.num_segments == 2
        ora2 = orc.Oracle(d)
        try:
            assert_same(eng.search_batch(queries, 10), ora2.search_batch(queries, 10), queries, "after reload onto two segments")
        finally:
            ora2.close()
    finally:
        eng.close()
        ora.close()


def test_damaged_frequent_lexicon_record_does_not_fail_the_load(index_factory, tmp_path):
    """Skip tables are an accelerator, not a load precondition (round-2 advice): the reference loads a segment whose
    lexicon holds a record pointing outside its inverted file (src/api_segment.cpp:83-100 checks nothing) and only the
    queries naming that term go wrong.  Here: the offset of the most frequent term of one barrel is pushed past the
    payload; reload() must succeed, every query that does not name the term must equal the oracle over the SAME damaged
    index (it never reads that record either), and a query naming it is refused at prepare with a message."""
    import shutil
    import struct
    src, _ = index_factory(2, 20_000, 4096, 77, False)
    d = str(tmp_path / "index")
    shutil.copytree(src, d)
    lex = os.path.join(d, "segments", "seg_000001", "lexicon_b000.bin")
    with open(lex, "rb") as f:
        blob = bytearray(f.read())
    (n,) = struct.unpack_from("<I", blob, 0)
    assert n > 0
    (ln,) = struct.unpack_from("<I", blob, 4)
    term = bytes(blob[8:8 + ln]).decode()
    at = 8 + ln                                   # termId u32, df u32, offset u64, count u32 (src/lexicon.cpp:116-120)
    df, = struct.unpack_from("<I", blob, at + 4)
    assert df >= 64                               # a frequent list: reload() registers it for a skip table
    struct.pack_into("<Q", blob, at + 8, 1 << 40)
    with open(lex, "wb") as f:
        f.write(blob)
    eng, ora = nsbind.Engine(d, 0), orc.Oracle(d)
    try:
        assert eng.num_segments == 2
        queries = [q for q in workloads.cfg5_queries(300, 5, 4096) + ["t000100 t000200", "zzzz"] if term not in q.split()]
        assert len(queries) > 100
        assert_same(eng.search_batch(queries, 10), ora.search_batch(queries, 10), queries, "damaged lexicon record, other terms")
        with pytest.raises(RuntimeError):
            eng.search_batch([term], 10)
    finally:
        eng.close()
        ora.close()


def test_and_extension_matches_derived_oracle(engines):
    g, eng, ora = engines("mid1")
    queries = workloads.cfg2_queries(64, 2002, g["params"]["vocab"]) + ["covid virus", "covid zzzzunknown", "covid covid virus"]
    gpu = eng.search_batch(queries, 10, nsbind.NS_FLAG_AND)
    assert_same(gpu, ora.search_batch(queries, 10, orc.FLAG_AND), queries, "AND")
    g8, eng8, ora8 = engines("multi8")
    assert_same(eng8.search_batch(queries, 10, nsbind.NS_FLAG_AND), ora8.search_batch(queries, 10, orc.FLAG_AND), queries, "AND multi8")


def test_json_surface(engines):
    import json

    g, eng, ora = engines("small2")
    j = json.loads(eng.search_json("covid virus", 5))
    assert list(j.keys()) == sorted(j.keys())   # nlohmann objects serialise alphabetically
    assert j["k"] == 5 and j["segments"] == 2 and j["query"] == "covid virus"
    oh, on, of, _ = ora.search_batch(["covid virus"], 5)
    assert j["found"] == int(of[0]) and len(j["results"]) == int(on[0])
    for r, o in zip(j["results"], oh[0]):
        assert list(r.keys()) == ["cord_uid", "docId", "score", "segment"]
        assert r["docId"] == int(o["doc"]) and r["segment"] == eng.segment_name(int(o["seg"]))
        assert np.float32(r["score"]) == o["score"]
        assert r["cord_uid"] == "u%08d" % (int(o["seg"]) * g["params"]["docs_per_segment"] + int(o["doc"]))
    early = json.loads(eng.search_json("the of", 10))
    assert "found" not in early and early["results"] == []
    assert json.loads(eng.search_json("covid", 1000))["k"] == 100   # clamp (src/api_engine.cpp:377)


def test_search_json_with_metadata_equals_reference_text(tmp_path_factory):
    """End to end on the device: Engine::search(query, k) -> JSON text must equal the REAL reference's
    dump(2) text (metadata decoration included) byte for byte, for every golden query whose returned
    scores are pairwise distinct (inside equal-score groups the reference's order is a hash-table
    artefact; those queries are compared as sets by the other golden tests)."""
    import json

    from test_host_cpu import _meta_index

    g, d, _ = _meta_index(tmp_path_factory)
    eng = nsbind.Engine(d, 0)
    exact = 0
    eng.set_cache(False)   # the golden texts are first answers (ref_driver empties the result cache per call)
    for case in g["cases"]:
        for q, text in zip(g["queries"], case["json"]):
            j = json.loads(text)
            scores = [r["score"] for r in j["results"]]
            mine = eng.search_json(q, case["k"])
            if len(set(scores)) == len(scores) and (not scores or j["found"] == len(scores) or True):
                jm = json.loads(mine)
                if [r["docId"] for r in jm["results"]] == [r["docId"] for r in j["results"]]:
                    assert mine == text, f"query {q!r} k={case['k']}"
                    exact += 1
                else:   # a tie at the K-th boundary resolved differently: ids may differ, the rest must not
                    assert jm["found"] == j["found"] and [r["score"] for r in jm["results"]] == scores
            else:
                jm = json.loads(mine)
                assert jm.get("found") == j.get("found") and sorted(r["score"] for r in jm["results"]) == sorted(scores)
    assert exact >= 12
    # the batch entry point assembles the same bodies (on several threads for large batches)
    qs = g["queries"] * 60
    bodies = eng.search_batch_json(qs, 5)
    assert len(bodies) == len(qs)
    for q, body in zip(qs[: len(g["queries"])], bodies):
        assert body == eng.search_json(q, 5)
    assert bodies[len(g["queries"]):2 * len(g["queries"])] == bodies[: len(g["queries"])]
    eng.close()


def test_search_cache_equals_reference_second_answers(index_factory):
    """Search-result cache around Engine::search (src/api_engine.cpp:190-250,:380-385,:539): the second answer to a
    query carries "from_cache": true, early returns are not cached, keys are "query|K"; then LRU eviction at 2600."""
    import json as _json
    with open(os.path.join(os.path.dirname(__file__), "golden", "cache1.json")) as f:
        g = _json.load(f)
    p = g["params"]
    d, _ = index_factory(p["n_segments"], p["docs_per_segment"], p["vocab"], p["seed"], p["legacy"])
    eng = nsbind.Engine(d, 0)
    try:
        for q, want in zip(g["queries"], g["second_answers"]):
            eng.search_json(q, g["k"])
            assert eng.search_json(q, g["k"]) == want, q
        assert eng.cache_size() == g["cache_entries"]
        assert '"from_cache"' not in eng.search_json(g["queries"][0], g["k"] + 1)      # another K, another key
        assert '"from_cache"' not in eng.search_json(g["queries"][0], 250)             # K is clamped to 100 before the key is made
        assert '"from_cache": true' in eng.search_json(g["queries"][0], 100)
        # eviction: 2600 entries, the least recently used one goes
        eng.set_cache(False); eng.set_cache(True)
        qs = [workloads.term_name(r) for r in range(9, 9 + 2601)]
        for q in qs[:2600]:
            eng.search_json(q, 1)
        assert eng.cache_size() == 2600
        assert '"from_cache": true' in eng.search_json(qs[0], 1)       # refreshes qs[0]; qs[1] is now the oldest
        eng.search_json(qs[2600], 1)                                     # evicts qs[1]
        assert eng.cache_size() == 2600
        assert '"from_cache"' not in eng.search_json(qs[1], 1)
        assert '"from_cache": true' in eng.search_json(qs[0], 1)
        eng.set_cache(False)
        assert '"from_cache"' not in eng.search_json(qs[0], 1) and eng.cache_size() == 0
    finally:
        eng.close()


def test_raw_abi_weights_and_errors(engines):
    """Direct ns_search_batch calls: fractional qweights (semantic-expansion shape) and argument errors."""
    g, eng, ora = engines("mid1")
    L = nsbind.hip_lib()
    a, b = eng.lookup(0, "covid"), eng.lookup(0, "t000050")
    refs = np.zeros(2, dtype=nsbind.TERM_DTYPE)
    refs[0] = (0, a["count"], a["byte_off"], a["idf"], 1.0)
    refs[1] = (0, b["count"], b["byte_off"], b["idf"], 0.6)   # alpha of SemanticIndex::expand
    qd = np.array([(0, 2)], dtype=nsbind.QDESC_DTYPE)
    rc, hits, nhits, found = nsbind.search_batch_raw(eng.ctx, qd, refs, 10)
    assert rc == 0
    # numpy restatement of the weighted sum with the oracle's per-term dense scores
    acc_a, t_a = ora.scores("covid", 0)
    acc_b, t_b = ora.scores("t000050", 0)
    acc = (np.where(t_a, acc_a, np.float32(0)) + np.float32(0.6) * np.where(t_b, acc_b, np.float32(0))).astype(np.float32)
    touched = t_a | t_b
    assert int(found[0]) == int(touched.sum())
    order = np.lexsort((np.arange(acc.size), -acc.astype(np.float64)))
    order = [d for d in order if touched[d]][:10]
    assert [int(d) for d in hits[0, : nhits[0]]["doc"]] == [int(d) for d in order]
    np.testing.assert_array_equal(hits[0, : nhits[0]]["score"].view(np.uint32), acc[order].view(np.uint32))
    # errors: K out of range, list outside the segment, unknown segment, misaligned offset
    assert nsbind.search_batch_raw(eng.ctx, qd, refs, 0)[0] == -1
    assert nsbind.search_batch_raw(eng.ctx, qd, refs, 101)[0] == -1
    bad = refs.copy(); bad[0]["count"] = 2**31
    assert nsbind.search_batch_raw(eng.ctx, qd, bad, 10)[0] == -1
    bad = refs.copy(); bad[0]["seg_id"] = 77
    assert nsbind.search_batch_raw(eng.ctx, qd, bad, 10)[0] == -1
    bad = refs.copy(); bad[0]["byte_off"] += 4
    assert nsbind.search_batch_raw(eng.ctx, qd, bad, 10)[0] == -1
    assert b"multiple of 8" in L.ns_last_error(eng.ctx)
    # empty batch and empty queries are fine
    rc, _, _, _ = nsbind.search_batch_raw(eng.ctx, np.zeros(0, nsbind.QDESC_DTYPE), np.zeros(0, nsbind.TERM_DTYPE), 10)
    assert rc == 0
    rc, hits, nhits, found = nsbind.search_batch_raw(eng.ctx, np.array([(0, 0), (0, 2)], dtype=nsbind.QDESC_DTYPE), refs, 10)
    assert rc == 0 and nhits[0] == 0 and found[0] == 0 and nhits[1] == 10
    # the padding the header promises — unused tail entries are {-inf, ~0, ~0} — for every K, also for a query without any
    # term ref next to queries with partial rows (round 3: such a query's row was only padded up to entry 63)
    for k in (1, 10, 64, 65, 100):
        rc, hits, nhits, found = nsbind.search_batch_raw(eng.ctx, np.array([(0, 0), (0, 2), (0, 0)], dtype=nsbind.QDESC_DTYPE), refs, k)
        assert rc == 0
        for q in range(3):
            tail = hits[q, nhits[q]:]
            assert np.all(tail["doc"] == 0xFFFFFFFF) and np.all(tail["seg"] == 0xFFFFFFFF) and np.all(np.isneginf(tail["score"])), (k, q)


def test_idf_outside_short_division_range_scales_exactly(engines):
    """The BM25 division takes its short form only when every idf of a group is in [2^-30, 2^30]
    (ns_div_short); outside it the full IEEE sequence runs.  Scaling every idf by 2^40 is exact in
    fp32 (no overflow here), so docs, order and `found` must be identical and every score exactly
    2^40 times the in-range one: the two division paths agree bit for bit."""
    g, eng, ora = engines("mid1")
    terms = ["covid", "t000050", "t000012", "t000300"]
    ent = [eng.lookup(0, t) for t in terms]
    qd = np.array([(0, 4), (4, 2), (6, 1)], dtype=nsbind.QDESC_DTYPE)
    order = [0, 1, 2, 3, 2, 0, 1]
    for scale in (2.0 ** 40, 2.0 ** -45):
        refs = np.zeros(len(order), dtype=nsbind.TERM_DTYPE)
        big = np.zeros(len(order), dtype=nsbind.TERM_DTYPE)
        for i, t in enumerate(order):
            e = ent[t]
            refs[i] = (0, e["count"], e["byte_off"], e["idf"], 1.0)
            big[i] = (0, e["count"], e["byte_off"], np.float32(e["idf"]) * np.float32(scale), 1.0)
        rc, h0, n0, f0 = nsbind.search_batch_raw(eng.ctx, qd, refs, 10)
        assert rc == 0
        rc, h1, n1, f1 = nsbind.search_batch_raw(eng.ctx, qd, big, 10)
        assert rc == 0
        np.testing.assert_array_equal(n0, n1)
        np.testing.assert_array_equal(f0, f1)
        for q in range(3):
            n = int(n0[q])
            np.testing.assert_array_equal(h0[q, :n]["doc"], h1[q, :n]["doc"])
            np.testing.assert_array_equal((h0[q, :n]["score"] * np.float32(scale)).view(np.uint32), h1[q, :n]["score"].view(np.uint32))


def _np_bm25(seg_lists, refs_idx, idfs, weights, doc_len, avgdl):
    """fp32 restatement of src/api_engine.cpp:477-480 in numpy (every operation rounds to fp32)."""
    f = np.float32
    acc = {}
    for li, idf, w in zip(refs_idx, idfs, weights):
        docs, tfs = seg_lists[li]
        dl = doc_len[docs].astype(np.float32)
        norm = f(1.2) * ((f(1.0) - f(0.75)) + f(0.75) * (dl / f(avgdl)))
        tf = tfs.astype(np.float32)
        s = (f(idf) * (tf * (f(1.2) + f(1.0)))) / (tf + norm)
        x = f(w) * s
        for d, v in zip(docs.tolist(), x.tolist()):
            acc[d] = f(acc.get(d, f(0.0)) + f(v))
    return acc


def test_sparse_lists_over_20m_docs_span_clamp():
    """A segment of 20 M docs with very sparse lists: one super-batch of the driver-stream body would
    span more docs than a table entry can identify (2^23), so the kernel clamps the span and walks
    through empty super-batches.  Checked against a numpy fp32 restatement through the raw C-ABI."""
    L = nsbind.hip_lib()
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    N = 20_000_000
    rng = np.random.default_rng(7)
    doc_len = rng.integers(20, 5000, size=N, dtype=np.uint32)
    avgdl = float(np.float32(doc_len.astype(np.float64).mean()))
    sizes = [700, 150, 40, 2500, 9]
    lists, payload = [], []
    for n in sizes:
        docs = np.sort(rng.choice(N, size=n, replace=False)).astype(np.uint32)
        tfs = rng.integers(1, 9, size=n, dtype=np.uint32)
        lists.append((docs, tfs))
        payload.append(np.stack([docs, tfs], axis=1).astype(np.uint32).ravel())
    # make lists 0 and 3 share docs so that accumulation across terms is exercised
    share = lists[3][0][::5][:100]
    docs0 = np.unique(np.concatenate([lists[0][0], share])).astype(np.uint32)
    lists[0] = (docs0, rng.integers(1, 9, size=len(docs0), dtype=np.uint32))
    payload[0] = np.stack(lists[0], axis=1).astype(np.uint32).ravel()
    flat = np.concatenate(payload)
    offs = np.cumsum([0] + [len(p) * 4 for p in payload])[:-1]
    seg = C.c_void_p()
    assert L.ns_segment_upload(ctx, 0, N, C.c_float(avgdl), doc_len.ctypes.data, flat.ctypes.data, flat.nbytes, C.byref(seg)) == 0, L.ns_last_error(ctx)
    idfs = [3.5, 5.25, 7.0, 2.125, 9.5]
    queries = [[0, 3], [3, 0, 1], [1, 2, 4], [4], [2, 0, 3, 1, 4, 0]]
    qd = np.zeros(len(queries), dtype=nsbind.QDESC_DTYPE)
    refs = []
    for qi, q in enumerate(queries):
        qd[qi] = (len(refs), len(q))
        for li in q:
            refs.append((0, len(lists[li][0]), int(offs[li]), idfs[li], 1.0 if li != 1 else 0.5))
    refs = np.array(refs, dtype=nsbind.TERM_DTYPE)
    for k in (10, 100):
        rc, hits, nhits, found = nsbind.search_batch_raw(ctx, qd, refs, k)
        assert rc == 0, L.ns_last_error(ctx)
        for qi, q in enumerate(queries):
            acc = _np_bm25(lists, q, [idfs[li] for li in q], [1.0 if li != 1 else 0.5 for li in q], doc_len, avgdl)
            assert int(found[qi]) == len(acc)
            order = sorted(acc.items(), key=lambda kv: (-float(kv[1]), kv[0]))[:k]
            n = int(nhits[qi])
            assert n == len(order)
            assert [int(d) for d in hits[qi, :n]["doc"]] == [d for d, _ in order]
            np.testing.assert_array_equal(hits[qi, :n]["score"].view(np.uint32), np.array([v for _, v in order], dtype=np.float32).view(np.uint32))
    assert L.ns_segment_release(ctx, seg) == 0
    L.ns_ctx_destroy(ctx)


def test_lone_query_wide_merge_ties_and_thresholds():
    """A lone query is cut into thousands of doc ranges and joined by k_merge_wide (threshold from the
    row heads, one sort).  Three shapes through the raw C-ABI against the numpy restatement: every
    score equal (mass ties: more candidates than the LDS buffer holds -> tournament fall-back), two
    score levels (ties AT the threshold), and distinct scores."""
    L = nsbind.hip_lib()
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    N = 600_000
    rng = np.random.default_rng(3)
    docs = np.arange(0, N, 2, dtype=np.uint32)
    shapes = {
        "all_equal": (np.full(N, 100, dtype=np.uint32), np.ones(len(docs), dtype=np.uint32)),
        "two_levels": (np.full(N, 100, dtype=np.uint32), np.where(np.arange(len(docs)) % 9973 == 5, 2, 1).astype(np.uint32)),
        "distinct": (rng.integers(20, 5000, size=N, dtype=np.uint32), rng.integers(1, 9, size=len(docs), dtype=np.uint32)),
    }
    for name, (doc_len, tfs) in shapes.items():
        avgdl = float(np.float32(doc_len.astype(np.float64).mean()))
        flat = np.stack([docs, tfs], axis=1).astype(np.uint32).ravel()
        seg = C.c_void_p()
        assert L.ns_segment_upload(ctx, 0, N, C.c_float(avgdl), doc_len.ctypes.data, flat.ctypes.data, flat.nbytes, C.byref(seg)) == 0, L.ns_last_error(ctx)
        qd = np.zeros(1, dtype=nsbind.QDESC_DTYPE)
        qd[0] = (0, 1)
        refs = np.array([(0, len(docs), 0, 1.75, 1.0)], dtype=nsbind.TERM_DTYPE)
        acc = _np_bm25([(docs, tfs)], [0], [1.75], [1.0], doc_len, avgdl)
        full = sorted(acc.items(), key=lambda kv: (-float(kv[1]), kv[0]))[:100]
        for min_items in (0, 20000):
            assert L.ns_set_tuning(ctx, 0, min_items, 0) == 0
            for k in (1, 10, 64, 100):
                rc, hits, nhits, found = nsbind.search_batch_raw(ctx, qd, refs, k)
                assert rc == 0, L.ns_last_error(ctx)
                order = full[:k]
                assert int(found[0]) == len(acc) and int(nhits[0]) == len(order), (name, k)
                assert [int(d) for d in hits[0, :len(order)]["doc"]] == [d for d, _ in order], (name, k, min_items)
                np.testing.assert_array_equal(hits[0, :len(order)]["score"].view(np.uint32), np.array([v for _, v in order], dtype=np.float32).view(np.uint32))
        assert L.ns_set_tuning(ctx, 0, 0, 0) == 0
        assert L.ns_segment_release(ctx, seg) == 0
    L.ns_ctx_destroy(ctx)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_impact_stream_equals_oracle_and_reference_golden(name, golden_index):
    """The optional second posting stream ({docId, precomputed term score}; ns_segment_build_impacts) must
    change nothing: same bytes as the oracle and as the real reference's captured output, in OR and AND
    mode, for every golden index; and the batches must really have read it (NS_INFO_IMPACTS)."""
    g, d, _ = golden_index(name)
    eng, ora = nsbind.Engine(d, 0), orc.Oracle(d)
    try:
        queries = g["queries"]
        plain = {k: eng.search_batch(queries, k) for k in (1, 10, 100)}
        b = eng.prepare(queries, 10)
        assert not (b.info().flags & nsbind.NS_INFO_IMPACTS)
        b.close()
        eng.build_impacts()
        b = eng.prepare(queries, 10)
        assert b.info().flags & nsbind.NS_INFO_IMPACTS
        b.close()
        for k in (1, 10, 100):
            gpu = eng.search_batch(queries, k)
            assert_same(gpu, ora.search_batch(queries, k), queries, f"{name} impacts k={k}")
            for a, c in zip(gpu, plain[k]):
                assert a.tobytes() == c.tobytes()
        assert_same(eng.search_batch(queries, 10, nsbind.NS_FLAG_AND), ora.search_batch(queries, 10, nsbind.NS_FLAG_AND), queries, f"{name} impacts AND")
        for case in g["cases"]:
            gh, gn, gf, gu = eng.search_batch(queries, case["k"])
            for qi, ref in enumerate(case["results"]):
                if ref["found"] < 0:
                    continue
                assert int(gf[qi]) == ref["found"]
                assert [int(x) for x in gh[qi, : gn[qi]]["score"].view(np.uint32)] == [h[2] for h in ref["hits"]]
        # lone queries and forced fine splitting go through the same bodies
        eng.set_tuning(0, 4096, 300)
        assert_same(eng.search_batch(queries[:20], 10), ora.search_batch(queries[:20], 10), queries[:20], f"{name} impacts split")
        eng.set_tuning(0, 0, 0)
        # switched off: back to the {docId, tf} stream
        eng.use_impacts(False)
        b = eng.prepare(queries, 10)
        assert not (b.info().flags & nsbind.NS_INFO_IMPACTS)
        b.close()
    finally:
        eng.close()
        ora.close()


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_packed_stream_equals_oracle_and_reference_golden(name, golden_index):
    """the compressed, blocked posting stream (ns_segment_build_packed: 256-posting blocks, 8/16/32-bit docId
    offsets, 8-bit tf, 16-bit index into the table of distinct norms) must change nothing: same bytes as the oracle, as
    the raw stream's answers and as the real reference's captured scores — OR and AND, K = 1/10/100, forced fine splits
    (cursors land inside blocks), alone and together with the impact stream — and the batches must really read it."""
    g, d, _ = golden_index(name)
    eng, ora = nsbind.Engine(d, 0), orc.Oracle(d)
    try:
        queries = g["queries"]
        plain = {k: eng.search_batch(queries, k) for k in (1, 10, 100)}
        plain_and = eng.search_batch(queries, 10, nsbind.NS_FLAG_AND)
        b = eng.prepare(queries, 10)
        assert not (b.info().flags & nsbind.NS_INFO_PACKED)
        b.close()
        eng.build_packed()
        for with_impacts, mode in ((False, 1), (False, 2), (True, 1)):
            if with_impacts:
                eng.build_impacts()
            eng.use_packed(mode)
            b = eng.prepare(queries, 10)
            fl = b.info().flags
            assert fl & nsbind.NS_INFO_PACKED and bool(fl & nsbind.NS_INFO_IMPACTS) == with_impacts
            b.close()
            for k in (1, 10, 100):
                gpu = eng.search_batch(queries, k)
                assert_same(gpu, ora.search_batch(queries, k), queries, f"{name} packed k={k} impacts={with_impacts}")
                for a, c in zip(gpu, plain[k]):
                    assert a.tobytes() == c.tobytes()
            for a, c in zip(eng.search_batch(queries, 10, nsbind.NS_FLAG_AND), plain_and):
                assert a.tobytes() == c.tobytes()
            for case in g["cases"]:
                gh, gn, gf, gu = eng.search_batch(queries, case["k"])
                for qi, ref in enumerate(case["results"]):
                    if ref["found"] < 0:
                        continue
                    assert int(gf[qi]) == ref["found"]
                    assert [int(x) for x in gh[qi, : gn[qi]]["score"].view(np.uint32)] == [h[2] for h in ref["hits"]]
            for tune in ((0, 4096, 300), (0, 1, 1 << 30), (0, 20000, 700)):
                eng.set_tuning(*tune)
                assert_same(eng.search_batch(queries[:24], 10), ora.search_batch(queries[:24], 10), queries[:24], f"{name} packed split {tune}")
            eng.set_tuning(0, 0, 0)
        eng.use_packed(0)
        b = eng.prepare(queries, 10)
        assert not (b.info().flags & nsbind.NS_INFO_PACKED)
        b.close()
    finally:
        eng.close()
        ora.close()


def test_packed_stream_escapes_and_limits():
    """Raw C-ABI on a crafted segment: tf values of 254, 255, 256 and 70000 (the tf plane holds 8 bits; 255 is the escape
    to the raw stream), lists that start and end in the middle of blocks, a block that straddles three lists (docIds not
    ascending inside it: 32-bit doc plane), dense runs (8-bit plane) next to sparse ones (16-bit), and a segment with more
    than 65536 distinct document lengths, for which the packed stream must be refused and the raw path keep working."""
    L = nsbind.hip_lib()
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    try:
        N = 300_000
        rng = np.random.default_rng(256)
        doc_len = rng.integers(20, 3000, size=N, dtype=np.uint32)
        avgdl = float(np.float32(doc_len.astype(np.float64).mean()))
        specs = [("dense", np.arange(1000, 1000 + 700)), ("sparse", np.sort(rng.choice(N, 900, replace=False))), ("tiny", np.array([5, 70000, 299999])),
                 ("mid", np.sort(rng.choice(60000, 5000, replace=False))), ("tiny2", np.array([9])), ("wide", np.sort(rng.choice(N, 300, replace=False)))]
        lists, payload = [], []
        for i, (_, docs) in enumerate(specs):
            docs = docs.astype(np.uint32)
            tfs = rng.integers(1, 12, size=len(docs), dtype=np.uint32)
            if len(docs) > 100:
                tfs[3] = 254; tfs[40] = 255; tfs[41] = 256; tfs[77] = 70000; tfs[len(docs) - 1] = 255
            lists.append((docs, tfs))
            payload.append(np.stack([docs, tfs], axis=1).astype(np.uint32).ravel())
        flat = np.concatenate(payload)
        offs = np.cumsum([0] + [len(p) * 4 for p in payload])[:-1]
        seg = C.c_void_p()
        assert L.ns_segment_upload(ctx, 0, N, C.c_float(avgdl), doc_len.ctypes.data, flat.ctypes.data, flat.nbytes, C.byref(seg)) == 0, L.ns_last_error(ctx)
        idfs = [1.25, 2.5, 6.0, 1.75, 9.0, 3.5]
        queries = [[0], [1], [3], [0, 3], [3, 0, 1], [2, 4, 5], [5, 1, 3, 0], [4], [3, 3], [1, 5]]
        qd = np.zeros(len(queries), dtype=nsbind.QDESC_DTYPE)
        refs = []
        for qi, q in enumerate(queries):
            qd[qi] = (len(refs), len(q))
            for li in q:
                refs.append((0, len(lists[li][0]), int(offs[li]), idfs[li], 1.0))
        refs = np.array(refs, dtype=nsbind.TERM_DTYPE)
        raw = {k: nsbind.search_batch_raw(ctx, qd, refs, k) for k in (10, 100)}
        assert L.ns_segment_build_packed(ctx, seg) == 0, L.ns_last_error(ctx)
        for split, mode in ((0, 1), (0, 2), (200, 1), (200, 2), (1 << 30, 2)):
            assert L.ns_set_tuning(ctx, 0, 1 if split else 0, split) == 0
            assert L.ns_ctx_use_packed(ctx, mode) == 0
            for k in (10, 100):
                b = nsbind.prepare_raw(ctx, qd, refs, k)
                assert b.info().flags & nsbind.NS_INFO_PACKED
                b.run(); hits, nhits, found = b.fetch(); b.close()
                for qi, q in enumerate(queries):
                    acc = _np_bm25(lists, q, [idfs[li] for li in q], [1.0] * len(q), doc_len, avgdl)
                    assert int(found[qi]) == len(acc)
                    order = sorted(acc.items(), key=lambda kv: (-float(kv[1]), kv[0]))[:k]
                    n = int(nhits[qi])
                    assert n == len(order) and [int(x) for x in hits[qi, :n]["doc"]] == [dd for dd, _ in order], (split, k, qi)
                    np.testing.assert_array_equal(hits[qi, :n]["score"].view(np.uint32), np.array([v for _, v in order], dtype=np.float32).view(np.uint32))
                if split == 0:
                    for a, c in zip((hits, nhits, found), raw[k][1:]):
                        assert a.tobytes() == c.tobytes()
        L.ns_set_tuning(ctx, 0, 0, 0)
        assert L.ns_segment_release(ctx, seg) == 0
        # more than 65536 distinct document lengths: no 16-bit norm index, no packed stream; the raw path is unaffected
        N2 = 70_000
        dl2 = (np.arange(N2, dtype=np.uint32) + 10)
        seg2 = C.c_void_p()
        small = np.array([[3, 1], [69_999, 2]], dtype=np.uint32).ravel()
        assert L.ns_segment_upload(ctx, 1, N2, C.c_float(float(np.float32(dl2.astype(np.float64).mean()))), dl2.ctypes.data, small.ctypes.data, small.nbytes, C.byref(seg2)) == 0
        assert L.ns_segment_build_packed(ctx, seg2) == -1 and b"65536 distinct" in L.ns_last_error(ctx)
        r2 = np.array([(1, 2, 0, 2.0, 1.0)], dtype=nsbind.TERM_DTYPE)
        b = nsbind.prepare_raw(ctx, np.array([(0, 1)], dtype=nsbind.QDESC_DTYPE), r2, 10)
        assert not (b.info().flags & nsbind.NS_INFO_PACKED)
        b.run(); h, n, f = b.fetch(); b.close()
        assert int(n[0]) == 2 and int(f[0]) == 2
        assert L.ns_segment_release(ctx, seg2) == 0
    finally:
        L.ns_ctx_destroy(ctx)


def test_impact_stream_partial_registration_and_foreign_idf():
    """Raw C-ABI: a batch reads impact streams only if EVERY list in it is registered with the bit-identical
    idf; otherwise it takes the {docId, tf} path.  Either way the numpy restatement must be met."""
    L = nsbind.hip_lib()
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    N = 200_000
    rng = np.random.default_rng(21)
    doc_len = rng.integers(20, 3000, size=N, dtype=np.uint32)
    avgdl = float(np.float32(doc_len.astype(np.float64).mean()))
    sizes = [60_000, 9_000, 300, 120_000, 17]
    lists, payload = [], []
    for n in sizes:
        docs = np.sort(rng.choice(N, size=n, replace=False)).astype(np.uint32)
        tfs = rng.integers(1, 30, size=n, dtype=np.uint32)
        lists.append((docs, tfs))
        payload.append(np.stack([docs, tfs], axis=1).astype(np.uint32).ravel())
    flat = np.concatenate(payload)
    offs = np.cumsum([0] + [len(p) * 4 for p in payload])[:-1].astype(np.uint64)
    seg = C.c_void_p()
    assert L.ns_segment_upload(ctx, 0, N, C.c_float(avgdl), doc_len.ctypes.data, flat.ctypes.data, flat.nbytes, C.byref(seg)) == 0, L.ns_last_error(ctx)
    idfs = np.array([1.5, 3.25, 6.0, 0.75, 9.5], dtype=np.float32)
    weights = [1.0, 0.5, 1.0, 1.0, 0.25]
    queries = [[0, 3], [3, 0, 1], [1, 2, 4], [4], [2, 0, 3, 1, 4, 0], [3], [0, 1]]

    def run(idf_of, expect_imp):
        qd = np.zeros(len(queries), dtype=nsbind.QDESC_DTYPE)
        refs = []
        for qi, q in enumerate(queries):
            qd[qi] = (len(refs), len(q))
            for li in q:
                refs.append((0, len(lists[li][0]), int(offs[li]), idf_of[li], weights[li]))
        refs = np.array(refs, dtype=nsbind.TERM_DTYPE)
        bh = C.c_void_p()
        assert L.ns_batch_prepare(ctx, qd.ctypes.data, refs.ctypes.data, len(qd), 10, 0, C.byref(bh)) == 0
        info = nsbind.NsBatchInfo()
        assert L.ns_batch_get_info(bh, C.byref(info)) == 0
        L.ns_batch_destroy(bh)
        assert bool(info.flags & nsbind.NS_INFO_IMPACTS) == expect_imp
        for k in (10, 100):
            rc, hits, nhits, found = nsbind.search_batch_raw(ctx, qd, refs, k)
            assert rc == 0, L.ns_last_error(ctx)
            for qi, q in enumerate(queries):
                acc = _np_bm25(lists, q, [idf_of[li] for li in q], [weights[li] for li in q], doc_len, avgdl)
                order = sorted(acc.items(), key=lambda kv: (-float(kv[1]), kv[0]))[:k]
                n = int(nhits[qi])
                assert int(found[qi]) == len(acc) and n == len(order)
                assert [int(x) for x in hits[qi, :n]["doc"]] == [x for x, _ in order], (qi, k, expect_imp)
                np.testing.assert_array_equal(hits[qi, :n]["score"].view(np.uint32), np.array([v for _, v in order], dtype=np.float32).view(np.uint32))

    cnts = np.array(sizes, dtype=np.uint32)
    run(idfs, False)
    # four of the five lists: a batch that touches list 4 cannot use the stream
    assert L.ns_segment_build_impacts(ctx, seg, offs[:4].ctypes.data, cnts[:4].ctypes.data, idfs[:4].ctypes.data, 4) == 0, L.ns_last_error(ctx)
    run(idfs, False)
    assert L.ns_segment_build_impacts(ctx, seg, offs[4:].ctypes.data, cnts[4:].ctypes.data, idfs[4:].ctypes.data, 1) == 0, L.ns_last_error(ctx)
    run(idfs, True)
    other = idfs.copy()
    other[3] = np.float32(0.8125)   # a caller-chosen idf the stream was not built with
    run(other, False)
    # re-registering the list with the new idf replaces its scores
    assert L.ns_segment_build_impacts(ctx, seg, offs[3:4].ctypes.data, cnts[3:4].ctypes.data, other[3:4].ctypes.data, 1) == 0
    run(other, True)
    run(idfs, False)
    # overlapping lists and lists outside the segment are rejected
    bad_off = np.array([offs[0], offs[0] + 8], dtype=np.uint64)
    bad_cnt = np.array([10, 10], dtype=np.uint32)
    assert L.ns_segment_build_impacts(ctx, seg, bad_off.ctypes.data, bad_cnt.ctypes.data, idfs[:2].ctypes.data, 2) != 0
    big = np.array([flat.nbytes - 8], dtype=np.uint64)
    assert L.ns_segment_build_impacts(ctx, seg, big.ctypes.data, bad_cnt[:1].ctypes.data, idfs[:1].ctypes.data, 1) != 0
    assert L.ns_segment_release(ctx, seg) == 0
    L.ns_ctx_destroy(ctx)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_shared_term_scores_equal_in_place_oracle_and_reference(name, golden_index):
    """ns_ctx_share_scores(2): the batch computes the BM25 term scores of its distinct lists once per run and scores from
    them.  Same bytes as scoring every posting in place, as the oracle and as the real reference's captured output, in OR and
    AND mode, at every K, under forced fine splitting and when the batch runs twice."""
    g, d, _ = golden_index(name)
    eng, ora = nsbind.Engine(d, 0), orc.Oracle(d)
    try:
        queries = g["queries"]
        eng.share_scores(0)
        plain = {k: eng.search_batch(queries, k) for k in (1, 10, 100)}
        b = eng.prepare(queries, 10)
        assert not (b.info().flags & (nsbind.NS_INFO_SHARED | nsbind.NS_INFO_IMPACTS))
        b.close()
        eng.share_scores(1)   # a golden batch is far too small to share by itself
        b = eng.prepare(queries, 10)
        assert not (b.info().flags & nsbind.NS_INFO_SHARED)
        b.close()
        eng.share_scores(2)
        b = eng.prepare(queries, 10)
        inf = b.info()
        assert (inf.flags & nsbind.NS_INFO_SHARED) and 0 < inf.shared_lists <= inf.n_term_refs and 0 < inf.shared_postings <= inf.postings
        first = None
        for _ in range(2):   # a second run builds the scores again
            b.run(False)
            got = b.fetch()
            first = first or got
            for a, c in zip(got, first):
                assert a.tobytes() == c.tobytes()
        b.close()
        for k in (1, 10, 100):
            gpu = eng.search_batch(queries, k)
            assert_same(gpu, ora.search_batch(queries, k), queries, f"{name} shared k={k}")
            for a, c in zip(gpu, plain[k]):
                assert a.tobytes() == c.tobytes()
        assert_same(eng.search_batch(queries, 10, nsbind.NS_FLAG_AND), ora.search_batch(queries, 10, nsbind.NS_FLAG_AND), queries, f"{name} shared AND")
        for case in g["cases"]:
            gh, gn, gf, gu = eng.search_batch(queries, case["k"])
            for qi, ref in enumerate(case["results"]):
                if ref["found"] < 0:
                    continue
                assert int(gf[qi]) == ref["found"]
                assert [int(x) for x in gh[qi, : gn[qi]]["score"].view(np.uint32)] == [h[2] for h in ref["hits"]]
        eng.set_tuning(0, 4096, 300)
        assert_same(eng.search_batch(queries[:20], 10), ora.search_batch(queries[:20], 10), queries[:20], f"{name} shared split")
        eng.set_tuning(0, 0, 0)
    finally:
        eng.close()
        ora.close()


def test_shared_term_scores_refusals_and_live_batches():
    """Raw C-ABI: when a batch may NOT share (ns_ctx_share_scores) — the same list under two idfs, a list overlapping another
    one, an idf a live sharing batch does not use, a segment whose optional impact stream lacks a list — it scores in place;
    either way the numpy restatement must be met, also by a live batch that runs again after others built into the buffer."""
    L = nsbind.hip_lib()
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    assert L.ns_ctx_share_scores(ctx, 3) != 0 and L.ns_ctx_share_scores(ctx, 2) == 0
    N = 150_000
    rng = np.random.default_rng(33)
    doc_len = rng.integers(20, 3000, size=N, dtype=np.uint32)
    avgdl = float(np.float32(doc_len.astype(np.float64).mean()))
    sizes = [50_000, 8_000, 400, 90_000, 23]
    lists, payload = [], []
    for n in sizes:
        docs = np.sort(rng.choice(N, size=n, replace=False)).astype(np.uint32)
        tfs = rng.integers(1, 30, size=n, dtype=np.uint32)
        lists.append((docs, tfs))
        payload.append(np.stack([docs, tfs], axis=1).astype(np.uint32).ravel())
    flat = np.concatenate(payload)
    offs = [int(x) for x in np.cumsum([0] + [len(p) * 4 for p in payload])[:-1]]
    # list 5: the second half of list 3 named as a list of its own (it overlaps list 3); list 6: list 0's start, shorter count
    lists.append((lists[3][0][45_000:], lists[3][1][45_000:])); offs.append(offs[3] + 45_000 * 8); sizes.append(45_000)
    lists.append((lists[0][0][:20_000], lists[0][1][:20_000])); offs.append(offs[0]); sizes.append(20_000)
    seg = C.c_void_p()
    assert L.ns_segment_upload(ctx, 0, N, C.c_float(avgdl), doc_len.ctypes.data, flat.ctypes.data, flat.nbytes, C.byref(seg)) == 0, L.ns_last_error(ctx)
    idfs = [1.5, 3.25, 6.0, 0.75, 9.5, 0.75, 1.5]
    weights = [1.0, 0.5, 1.0, 1.0, 0.25, 1.0, 1.0]

    def descs(queries, idf_of):
        qd = np.zeros(len(queries), dtype=nsbind.QDESC_DTYPE)
        refs = []
        for qi, q in enumerate(queries):
            qd[qi] = (len(refs), len(q))
            for li in q:
                li, idf = li if isinstance(li, tuple) else (li, idf_of[li])
                refs.append((0, sizes[li], offs[li], idf, weights[li]))
        return qd, np.array(refs, dtype=nsbind.TERM_DTYPE)

    def check(queries, idf_of, hits, nhits, found, k, label):
        for qi, q in enumerate(queries):
            ids = [li[0] if isinstance(li, tuple) else li for li in q]
            fs = [li[1] if isinstance(li, tuple) else idf_of[li] for li in q]
            acc = _np_bm25(lists, ids, fs, [weights[i] for i in ids], doc_len, avgdl)
            order = sorted(acc.items(), key=lambda kv: (-float(kv[1]), kv[0]))[:k]
            n = int(nhits[qi])
            assert int(found[qi]) == len(acc) and n == len(order), (label, qi)
            assert [int(x) for x in hits[qi, :n]["doc"]] == [x for x, _ in order], (label, qi)
            np.testing.assert_array_equal(hits[qi, :n]["score"].view(np.uint32), np.array([v for _, v in order], dtype=np.float32).view(np.uint32))

    def prepare(queries, idf_of, expect_shared, k=10):
        qd, refs = descs(queries, idf_of)
        bh = C.c_void_p()
        assert L.ns_batch_prepare(ctx, qd.ctypes.data, refs.ctypes.data, len(qd), k, 0, C.byref(bh)) == 0, L.ns_last_error(ctx)
        info = nsbind.NsBatchInfo()
        assert L.ns_batch_get_info(bh, C.byref(info)) == 0
        assert bool(info.flags & nsbind.NS_INFO_SHARED) == expect_shared, (queries, info.flags)
        return bh, (qd, refs)

    def run_fetch(bh, queries, idf_of, label, k=10):
        Q = len(queries)
        hits = np.empty((Q, k), dtype=nsbind.HIT_DTYPE); nhits = np.zeros(Q, np.uint32); found = np.zeros(Q, np.uint64)
        assert L.ns_batch_run(bh, 0) == 0, L.ns_last_error(ctx)
        assert L.ns_batch_fetch(bh, hits.ctypes.data, nhits.ctypes.data, found.ctypes.data) == 0, L.ns_last_error(ctx)
        check(queries, idf_of, hits, nhits, found, k, label)

    base = [[0, 3], [3, 0, 1], [1, 2, 4], [4], [2, 0, 3, 1, 4, 0], [3], [0, 1]]
    b1, _ = prepare(base, idfs, True)
    run_fetch(b1, base, idfs, "shared")
    # the same list under two idfs inside one batch: no sharing
    two = [[0, (0, 2.0)], [3, 1]]
    b2, _ = prepare(two, idfs, False)
    run_fetch(b2, two, idfs, "two idfs in one batch")
    L.ns_batch_destroy(b2)
    # another idf for list 3 while b1 (which built it with 0.75) is alive: no sharing; b1 still right when it runs again
    other = list(idfs); other[3] = 0.8125
    b3, _ = prepare(base, other, False)
    run_fetch(b3, base, other, "foreign idf, live batch")
    run_fetch(b1, base, idfs, "live batch again")
    # a second live sharing batch with the same idfs builds into the same buffer; both stay right in any order
    more = [[1, 0], [2, 3, 4], [0]]
    b4, _ = prepare(more, idfs, True)
    run_fetch(b4, more, idfs, "second live batch")
    run_fetch(b1, base, idfs, "first live batch after the second")
    # the optional stream cannot be built under live sharing batches
    o = np.array([offs[0]], np.uint64); c = np.array([sizes[0]], np.uint32); f = np.array([idfs[0]], np.float32)
    assert L.ns_segment_build_impacts(ctx, seg, o.ctypes.data, c.ctypes.data, f.ctypes.data, 1) != 0
    L.ns_batch_destroy(b3); L.ns_batch_destroy(b4); L.ns_batch_destroy(b1)
    # no live sharing batch left: the new idf takes the list over
    b5, _ = prepare(base, other, True)
    run_fetch(b5, base, other, "idf replaced")
    L.ns_batch_destroy(b5)
    b6, _ = prepare(base, idfs, True)
    run_fetch(b6, base, idfs, "idf back")
    L.ns_batch_destroy(b6)
    # lists that overlap a shared one (a tail of list 3, a shorter list 0) are refused, whichever batch names them
    for q in ([[5, 1], [3]], [[6], [1, 2]], [[5]]):
        bo, _ = prepare(q, idfs, False)
        run_fetch(bo, q, idfs, "overlapping list")
        L.ns_batch_destroy(bo)
    b7, _ = prepare(base, idfs, True)   # the refusals left the registry intact
    run_fetch(b7, base, idfs, "after refusals")
    L.ns_batch_destroy(b7)
    # K = 100 and the one-shot entry point
    qd, refs = descs(base, idfs)
    rc, hits, nhits, found = nsbind.search_batch_raw(ctx, qd, refs, 100)
    assert rc == 0
    check(base, idfs, hits, nhits, found, 100, "one-shot K=100")
    # a segment with an optional impact stream that lacks one of the batch's lists: in place; with all of them: that stream
    o = np.array(offs[:4], np.uint64); c = np.array(sizes[:4], np.uint32); f = np.array(idfs[:4], np.float32)
    assert L.ns_segment_build_impacts(ctx, seg, o.ctypes.data, c.ctypes.data, f.ctypes.data, 4) == 0, L.ns_last_error(ctx)
    b8, _ = prepare(base, idfs, False)
    info = nsbind.NsBatchInfo(); L.ns_batch_get_info(b8, C.byref(info))
    assert not (info.flags & nsbind.NS_INFO_IMPACTS)
    run_fetch(b8, base, idfs, "optional stream lacks a list")
    L.ns_batch_destroy(b8)
    part = [[0, 3], [3, 0, 1], [1]]
    b9, _ = prepare(part, idfs, False)
    info = nsbind.NsBatchInfo(); L.ns_batch_get_info(b9, C.byref(info))
    assert info.flags & nsbind.NS_INFO_IMPACTS
    run_fetch(b9, part, idfs, "optional stream")
    L.ns_batch_destroy(b9)
    assert L.ns_segment_release(ctx, seg) == 0
    # a NEW segment under the released segment's id: the registry has forgotten the old lists (list 7 starts where list 0 did, with
    # another count: it is a new list, admitted and shared), and overlaps are checked afresh (list 8 overlaps list 7: refused)
    lists.append((lists[0][0][:30_000], lists[0][1][:30_000])); offs.append(offs[0]); sizes.append(30_000); idfs.append(2.5); weights.append(1.0)
    lists.append((lists[0][0][10_000:40_000], lists[0][1][10_000:40_000])); offs.append(offs[0] + 10_000 * 8); sizes.append(30_000); idfs.append(2.5); weights.append(1.0)
    seg2 = C.c_void_p()
    assert L.ns_segment_upload(ctx, 0, N, C.c_float(avgdl), doc_len.ctypes.data, flat.ctypes.data, flat.nbytes, C.byref(seg2)) == 0, L.ns_last_error(ctx)
    fresh = [[7, 1], [7], [2, 7, 4]]
    b10, _ = prepare(fresh, idfs, True)
    run_fetch(b10, fresh, idfs, "new segment under the old id")
    L.ns_batch_destroy(b10)
    over = [[8, 1], [7]]
    b11, _ = prepare(over, idfs, False)
    run_fetch(b11, over, idfs, "overlap in the new segment")
    L.ns_batch_destroy(b11)
    assert L.ns_segment_release(ctx, seg2) == 0
    L.ns_ctx_destroy(ctx)


def test_many_terms_per_query(engines):
    """More scored terms than one wave-pass handles (64) and than the reference's expansion cap (40)."""
    g, eng, ora = engines("mid1")
    q70 = " ".join(workloads.term_name(r) for r in range(3, 73))
    q130 = " ".join(workloads.term_name(1 + (r * 7) % 300) for r in range(130))
    q12 = " ".join(workloads.term_name(r) for r in range(2, 14))      # 9..16 terms: binary-search term lookup, narrow tables
    q20 = " ".join(workloads.term_name(5 + 3 * r) for r in range(20))   # 17..64 terms: the wide instantiation
    q40 = " ".join(workloads.term_name(1 + (r * 11) % 200) for r in range(40))
    queries = [q70, q130, "covid " * 20, q12, q20, q40, q12 + " covid covid"]
    assert_same(eng.search_batch(queries, 100), ora.search_batch(queries, 100), queries, "many terms")


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4", "cfg5"])
def test_config_shaped_workloads_reduced(cfg, index_factory):
    """BASELINE configs at sizes the oracle finishes in seconds: same query laws, smaller index/batch."""
    gen, _, K, flags, (nseg, docs) = workloads.WORKLOADS[cfg]
    docs_small = {"cfg2": 100_000, "cfg3": 60_000, "cfg4": 12_500, "cfg5": 60_000}[cfg]
    d, _ = index_factory(nseg, docs_small, 65536, 1337, False)
    queries = gen(256)
    eng, ora = nsbind.Engine(d, 0), orc.Oracle(d)
    try:
        assert_same(eng.search_batch(queries, K, flags), ora.search_batch(queries, K, flags), queries, cfg)
    finally:
        eng.close()
        ora.close()


def test_full_size_1m_docs_properties_and_sample(index_factory):
    """BASELINE full sizes (1M docs; cfg5 16384 queries, cfg3 4096 x K=100): size-independent
    properties on the whole batch + bit-exact oracle comparison on a sample."""
    d, total = index_factory(1, 1_000_000, 65536, 1337, False)
    eng, ora = nsbind.Engine(d, 0), orc.Oracle(d)
    try:
        for cfg, nsample in (("cfg5", 384), ("cfg3", 96)):
            gen, Q, K, flags, _ = workloads.WORKLOADS[cfg]
            queries = gen(Q)
            hits, nhits, found, usable = eng.search_batch(queries, K, flags)
            assert usable.all()
            valid = np.arange(K)[None, :] < nhits[:, None]
            s = np.where(valid, hits["score"], -np.inf)
            # sortedness: score non-increasing, ties by ascending doc id
            assert np.all(s[:, :-1] >= s[:, 1:])
            tie = valid[:, 1:] & (s[:, :-1] == s[:, 1:])
            assert np.all(hits["doc"][:, :-1][tie] < hits["doc"][:, 1:][tie])
            assert np.all(nhits == np.minimum(found, K))
            assert np.all(hits["doc"][valid] < 1_000_000) and np.all(hits["seg"][valid] == 0)
            # idempotence: a second pass over the same batch returns identical bytes
            h2, n2, f2, _ = eng.search_batch(queries, K, flags)
            assert hits.tobytes() == h2.tobytes() and np.array_equal(found, f2)
            # single-term queries: found == LexEntry.count
            singles = [i for i, q in enumerate(queries) if len(q.split()) == 1][:50]
            for i in singles:
                assert int(found[i]) == eng.lookup(0, queries[i])["count"]
            # sample against the oracle, bit for bit
            idx = np.linspace(0, Q - 1, nsample).astype(int)
            sub = [queries[i] for i in idx]
            assert_same((hits[idx], nhits[idx], found[idx], usable[idx]), ora.search_batch(sub, K, flags, threads=16), sub, cfg + " full")
            # the same queries one at a time (each spread over the whole chip, joined by k_merge_wide)
            for i in idx[:12]:
                lone = eng.search_batch([queries[i]], K, flags)
                assert_same(lone, (hits[i:i + 1], nhits[i:i + 1], found[i:i + 1], usable[i:i + 1]), [queries[i]], cfg + " lone")
        # the optional impact stream changes no byte of either full batch
        plain = {}
        for cfg in ("cfg5", "cfg3", "cfg2"):
            gen, Q, K, flags, _ = workloads.WORKLOADS[cfg]
            plain[cfg] = eng.search_batch(gen(Q), K, flags)
        eng.build_impacts()
        for cfg in ("cfg5", "cfg3", "cfg2"):
            gen, Q, K, flags, _ = workloads.WORKLOADS[cfg]
            b = eng.prepare(gen(Q), K, flags)
            assert b.info().flags & nsbind.NS_INFO_IMPACTS
            b.close()
            for a, c in zip(eng.search_batch(gen(Q), K, flags), plain[cfg]):
                assert a.tobytes() == c.tobytes(), cfg
    finally:
        eng.close()
        ora.close()


def test_multi_segment_8x125k_global_heap(index_factory):
    """cfg4 shape at full size: per-segment stats, one global top-K across segments."""
    d, _ = index_factory(8, 125_000, 65536, 1337, False)
    eng, ora = nsbind.Engine(d, 0), orc.Oracle(d)
    try:
        queries = workloads.cfg4_queries(512)
        gpu = eng.search_batch(queries, 10)
        idx = np.arange(0, 512, 8)
        sub = [queries[i] for i in idx]
        assert_same(tuple(a[idx] for a in gpu), ora.search_batch(sub, 10, threads=16), sub, "cfg4 full")
        assert len(set(int(x) for x in gpu[0]["seg"][gpu[0]["doc"] != 0xFFFFFFFF])) > 1
    finally:
        eng.close()
        ora.close()


def test_staged_batch_with_bound_torch_outputs(engines):
    """bench.py's path: descriptors resident, results written into torch device tensors."""
    torch = pytest.importorskip("torch")
    g, eng, ora = engines("mid1")
    queries = g["queries"]
    Q, K = len(queries), 10
    b = eng.prepare(queries, K)
    d_hits = torch.zeros((Q, K, 3), dtype=torch.int32, device="cuda")
    d_nhits = torch.zeros(Q, dtype=torch.int32, device="cuda")
    d_found = torch.zeros(Q, dtype=torch.int64, device="cuda")
    b.bind_outputs(d_hits.data_ptr(), d_nhits.data_ptr(), d_found.data_ptr())
    torch.cuda.synchronize()   # the ctx runs on its own non-blocking stream: finish torch's fills first
    b.run(timed=True)
    b.sync()
    info = b.info()
    assert info.last_score_kernel_ms > 0 and info.algo_bytes == 8 * info.postings
    hits = d_hits.cpu().numpy().view(np.uint32).reshape(Q, K, 3)
    oh, on, of, ou = ora.search_batch(queries, K)
    for q in range(Q):
        n = int(on[q]) if ou[q] else 0
        assert int(d_nhits[q]) == n
        if ou[q]:
            assert int(d_found[q]) == int(of[q])
        np.testing.assert_array_equal(hits[q, :n, 0], oh[q, :n]["score"].view(np.uint32))
        np.testing.assert_array_equal(hits[q, :n, 2], oh[q, :n]["doc"])
    b.close()

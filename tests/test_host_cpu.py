"""CPU-only tests: on-disk format, generator, host-side query preparation, and that the C-ABI
libraries load and export every symbol the headers declare.  No compute call is made here — the
product has no CPU scoring path, and the test below asserts that it fails loudly without a GPU."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

import nsbind
import workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    with open(os.path.join(ROOT, "include", header)) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


def test_abi_exports_every_declared_symbol():
    hip, host = nsbind.hip_lib(), nsbind.host_lib()
    declared = _declared("nextsearch_hip.h", "ns_")
    assert sorted(nsbind.HIP_SYMBOLS) == declared
    for s in declared:
        assert getattr(hip, s) is not None
    declared_h = _declared("nextsearch_host.h", "nsh_")
    assert sorted(nsbind.HOST_SYMBOLS) == declared_h
    for s in declared_h:
        assert getattr(host, s) is not None


def test_struct_layouts_match_header():
    assert C.sizeof(nsbind.NsTermRef) == 24 and nsbind.NsTermRef.byte_off.offset == 8
    assert C.sizeof(nsbind.NsHit) == 12 and C.sizeof(nsbind.NsQueryDesc) == 8
    assert nsbind.NsBatchInfo.sum_score_kernel_ms.offset % 8 == 0


def test_no_cpu_fallback(gpu_available, golden_index):
    """Without a device the product fails loudly: ctx creation errors out and the facade refuses to search."""
    if gpu_available:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = nsbind.hip_lib().ns_ctx_create(0, C.byref(h))
    assert rc == -2 and not h.value
    assert b"no CPU fallback" in nsbind.hip_lib().ns_last_error(None)
    _, d, _ = golden_index("small2")
    with pytest.raises(RuntimeError, match="ns_ctx_create"):
        nsbind.Engine(d, 0)
    eng = nsbind.Engine(d, -1)   # host-only: index + query prep
    with pytest.raises(RuntimeError, match="no device context"):
        eng.search_batch(["covid"], 10)
    eng.close()


def _read_str(b, pos):
    (n,) = struct.unpack_from("<I", b, pos)
    return b[pos + 4 : pos + 4 + n].decode(), pos + 4 + n


@pytest.mark.parametrize("name", ["small2", "legacy1"])
def test_loader_matches_independent_parse_of_files(name, golden_index):
    """Host loader (barrels flattened into one buffer + base table) vs a struct.unpack parse of the files."""
    g, d, total = golden_index(name)
    eng = nsbind.Engine(d, -1)
    p = g["params"]
    assert eng.num_segments == p["n_segments"]
    seen = 0
    for s in range(eng.num_segments):
        segdir = os.path.join(d, "segments", eng.segment_name(s))
        info = eng.segment_info(s)
        with open(os.path.join(segdir, "stats.bin"), "rb") as f:
            N, avgdl = struct.unpack("<If", f.read(8))
        assert (info["n_docs"], info["avgdl"]) == (N, avgdl) and N == p["docs_per_segment"]
        dl = eng.segment_doc_len(s)
        assert np.float32(np.float32(int(dl.astype(np.uint64).sum())) / np.float32(N)) == np.float32(avgdl)
        assert dl.min() >= 20 and dl.max() < 16384
        post = eng.segment_postings(s)
        seen += len(post)
        assert info["use_barrels"] == (not p["legacy"])
        lexfiles = ([("lexicon_b%03d.bin" % b, "inverted_b%03d.bin" % b) for b in range(64)] if not p["legacy"]
                    else [("lexicon.bin", "inverted.bin")])
        base = 0
        nterms = 0
        for lf, invf in lexfiles:
            with open(os.path.join(segdir, lf), "rb") as f:
                lb = f.read()
            with open(os.path.join(segdir, invf), "rb") as f:
                inv = np.frombuffer(f.read(), dtype="<u4").reshape(-1, 2)
            (tcount,) = struct.unpack_from("<I", lb, 0)
            pos = 4
            for _ in range(tcount):
                term, pos = _read_str(lb, pos)
                tid, df, off, cnt = struct.unpack_from("<IIQI", lb, pos)
                pos += 20
                nterms += 1
                e = eng.lookup(s, term)
                assert e is not None and (e["term_id"], e["df"], e["count"]) == (tid, df, cnt)
                assert e["byte_off"] == base + off and off % 8 == 0 and cnt == df
                lst = post[e["byte_off"] // 8 : e["byte_off"] // 8 + cnt]
                np.testing.assert_array_equal(lst, inv[off // 8 : off // 8 + cnt])
                assert np.all(np.diff(lst[:, 0].astype(np.int64)) > 0) and lst[:, 0].max() < N   # docId strictly ascending
                assert lst[:, 1].min() >= 1 and lst[:, 1].max() <= 64
            assert pos == len(lb)
            base += inv.size * 4
        assert nterms == info["n_terms"]
    assert seen == total == g["total_postings"]
    eng.close()


def test_tokenizer_and_base_terms():
    L = nsbind.host_lib()
    buf = C.create_string_buffer(256)

    def bt(q):
        n = L.nsh_base_terms(q if isinstance(q, bytes) else q.encode(), buf, 256)
        return n, buf.value.decode()

    assert bt("COVID-19: the Virus, of a vaccine!") == (4, "covid 19 virus vaccine")
    assert bt("covid covid") == (2, "covid covid")                 # duplicates kept
    assert bt("the a an and or of to in for on with by as is are was were be been it this that from at") == (0, "")
    assert bt("x y z 1 2") == (0, "")                               # size() < 2 dropped
    assert bt(b"caf\xc3\xa9 na\xefve") == (3, "caf na ve")          # bytes >= 0x80 split tokens
    assert bt("a1b2_c3  d4") == (3, "a1b2 c3 d4")
    assert bt("") == (0, "")


def test_idf_matches_reference_expression():
    """bm25_idf (src/api_engine.cpp:45-47) BIT FOR BIT: `N - df` is a u32 subtraction (it wraps when df > N), then
    int -> float, two fp32 additions, an fp32 division, and glibc's logf — restated here with numpy fp32 scalars and
    the C library's logf called through ctypes (the function the reference's std::log(float) resolves to)."""
    import ctypes
    import ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    libm.logf.argtypes = [ctypes.c_float]
    libm.logf.restype = ctypes.c_float
    L = nsbind.host_lib()
    f = np.float32
    rng = np.random.default_rng(45)
    cases = [(1000, 1), (1000, 999), (1000, 1000), (1_000_000, 600_000), (5, 9), (0, 0), (1, 0), (2**32 - 1, 1), (16_777_217, 3),
             (125_000, 74_907), (1_000_000, 599_412)]
    cases += [(int(n), int(d)) for n, d in zip(rng.integers(1, 2**31, 4000), rng.integers(0, 2**31, 4000))]
    cases += [(int(n), int(rng.integers(0, n + 1))) for n in rng.integers(1, 5_000_000, 4000)]
    for N, df in cases:
        num = f(np.uint32((N - df) & 0xFFFFFFFF)) + f(0.5)
        den = f(np.uint32(df)) + f(0.5)
        want = f(libm.logf(ctypes.c_float(float(f(f(num / den) + f(1.0))))))
        got = f(L.nsh_bm25_idf(N, df))
        assert got.view(np.uint32) == want.view(np.uint32), (N, df, float(got), float(want))


def test_build_refs_layout(golden_index):
    g, d, _ = golden_index("small2")
    eng = nsbind.Engine(d, -1)
    queries = ["covid virus", "the of", "zzzz", "virus covid covid"]
    qd, refs, usable = eng.build_refs(queries)
    assert list(usable) == [1, 0, 1, 1]
    assert list(qd["term_count"]) == [4, 0, 0, 6]    # 2 segments x terms, segment-major
    assert list(refs["seg_id"][:4]) == [0, 0, 1, 1]
    c0, v0 = eng.lookup(0, "covid"), eng.lookup(0, "virus")
    assert (refs[0]["byte_off"], refs[0]["count"]) == (c0["byte_off"], c0["count"])
    assert (refs[1]["byte_off"], refs[1]["count"]) == (v0["byte_off"], v0["count"])
    assert refs[0]["idf"] == np.float32(c0["idf"]) and np.all(refs["qweight"] == 1.0)
    # query-term order inside the segment group: virus, covid, covid
    off = qd["term_begin"][3]
    assert [int(x) for x in refs["byte_off"][off : off + 3]] == [v0["byte_off"], c0["byte_off"], c0["byte_off"]]
    eng.close()


def test_build_refs_threaded_equals_per_query(golden_index):
    """Batches of >= 1024 queries are prepared by several host threads on contiguous slices; the result
    must be the concatenation, in query order, of what each query gives on its own."""
    g, d, _ = golden_index("small2")
    eng = nsbind.Engine(d, -1)
    base = ["covid virus", "the of", "zzzz", "virus covid covid", "vaccine", "", "covid-19 in children"]
    queries = [base[(i * 7 + i // 5) % len(base)] for i in range(3000)]
    qd, refs, usable = eng.build_refs(queries)
    single = {q: eng.build_refs([q]) for q in set(queries)}
    pos = 0
    for i, q in enumerate(queries):
        sqd, srefs, su = single[q]
        assert usable[i] == su[0] and qd["term_count"][i] == sqd["term_count"][0]
        n = int(sqd["term_count"][0])
        if n:
            assert qd["term_begin"][i] == pos
            assert refs[pos : pos + n].tobytes() == srefs[:n].tobytes()
        pos += n
    assert pos == len(refs)
    eng.close()


def _meta_index(tmp_path_factory):
    """The meta1 fixture's index: generator output + the deterministic metadata.csv (own directory:
    the other golden tests expect undecorated results)."""
    import hashlib

    from conftest import load_golden

    g = load_golden("meta1")
    p = g["params"]
    d = str(tmp_path_factory.mktemp("idx_meta") / "index")
    nsbind.gen_index(d, p["n_segments"], p["docs_per_segment"], p["vocab"], p["seed"], p["legacy"])
    csv = workloads.metadata_csv(p["n_segments"] * p["docs_per_segment"], p["meta_seed"])
    assert hashlib.sha256(csv).hexdigest() == g["metadata_csv_sha256"]
    with open(os.path.join(d, "metadata.csv"), "wb") as f:
        f.write(csv)
    return g, d, csv


def test_result_assembly_equals_reference_json_text(tmp_path_factory):
    """decoration from metadata.csv + JSON layout.  The golden holds the REAL
    reference's Engine::search(...).dump(2) text; feeding its hits to this repo's result assembly must
    reproduce that text byte for byte (keys, indentation, escaping, float spelling, optional fields)."""
    import json

    g, d, _ = _meta_index(tmp_path_factory)
    eng = nsbind.Engine(d, -1)
    names = [eng.segment_name(s) for s in range(eng.num_segments)]
    n = 0
    for case in g["cases"]:
        for q, text in zip(g["queries"], case["json"]):
            j = json.loads(text)
            hits = np.zeros(len(j["results"]), dtype=nsbind.HIT_DTYPE)
            for i, r in enumerate(j["results"]):
                hits[i] = (np.float32(r["score"]), names.index(r["segment"]), r["docId"])
            mine = eng.hits_to_json(q, case["k"], "found" in j, j.get("found", 0), hits)
            assert mine == text, f"query {q!r} k={case['k']}"
            n += len(hits)
    assert n > 40
    eng.close()


@pytest.mark.parametrize("slice_bytes", [0, 40_000, 977])
def test_metadata_table_equals_python_restatement(tmp_path_factory, monkeypatch, slice_bytes):
    """Every document's decorated fields against an independent restatement of src/api_metadata.cpp's rules
    (physical lines, quote toggling without escapes, last header column wins, first row per cord_uid wins,
    url cut at ';', first_author_et_al).  slice_bytes > 0 forces the loader's several-threads path on the small
    fixture (slices cut at line ends, merged in file order: duplicates of a cord_uid may sit in different slices)."""
    g, d, csv = _meta_index(tmp_path_factory)
    if slice_bytes:
        monkeypatch.setenv("NS_META_SLICE_BYTES", str(slice_bytes))

    def split(line):
        out, cur, inq = [], [], False
        for ch in line:
            if ch == '"':
                inq = not inq
            elif ch == "," and not inq:
                out.append("".join(cur)); cur = []
            else:
                cur.append(ch)
        out.append("".join(cur))
        return out

    ws = " \t\n\r\f\v"

    def author(raw):
        s = raw.strip(ws)
        if not s:
            return ""
        first = s.split(";", 1)[0].strip(ws)
        while first and (first[-1] == "," or first[-1] in ws):
            first = first[:-1]
        first = first.strip(ws)
        if not first:
            return ""
        if first[0] == "(":
            c = first.find(")")
            if c > 1 and first[1:c].strip(ws):
                first = first[1:c].strip(ws)
        if "," in first:
            sur = first.split(",", 1)[0].strip(ws)
        else:
            t = first.strip(ws)
            k = max(t.rfind(" "), t.rfind("\t"))
            sur = t if k < 0 else t[k + 1:].strip(ws)
        return sur + " et al." if sur else ""

    text = csv.decode("utf-8")
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    cols = split(lines[0])
    idx = {name: max(i for i, c in enumerate(cols) if c == name) for name in ("cord_uid", "url", "publish_time", "authors", "title")}
    want = {}
    for line in lines[1:]:
        r = split(line)
        if len(r) <= idx["cord_uid"] or not r[idx["cord_uid"]] or r[idx["cord_uid"]] in want:
            continue
        get = lambda k: r[idx[k]] if len(r) > idx[k] else ""   # noqa: E731
        want[r[idx["cord_uid"]]] = {"title": get("title"), "url": get("url").split(";", 1)[0], "publish_time": get("publish_time"),
                                    "author": author(get("authors"))}
    eng = nsbind.Engine(d, -1)
    per = g["params"]["docs_per_segment"]
    checked = 0
    for s in range(eng.num_segments):
        for doc in range(per):
            uid = "u%08d" % (s * per + doc)
            assert eng.doc_metadata(s, doc) == want.get(uid), uid
            checked += uid in want
    assert checked > 3000
    eng.close()


def test_generator_shapes():
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        total = nsbind.gen_index(os.path.join(tmp, "i"), 1, 50_000, 4096, 1337, False)
        eng = nsbind.Engine(os.path.join(tmp, "i"), -1)
        n = 50_000
        top = eng.lookup(0, "covid")            # rank 1: df target 0.6 N (Bernoulli sweep)
        assert abs(top["count"] - 0.6 * n) < 0.02 * n
        r100 = eng.lookup(0, "t000100")         # sparse path: 0.6 N / 100 draws, minus collisions
        assert 0.9 * 0.006 * n < r100["count"] <= 0.006 * n
        assert eng.lookup(0, "t004096")["count"] >= 1
        assert eng.lookup(0, "t004097") is None
        assert abs(total - 0.6 * n * sum(1.0 / r for r in range(1, 4097))) < 0.03 * total
        eng.close()


def test_search_cache_holds_only_answers():
    """Search-result cache (src/api_engine.cpp:195-250), host side only: the engine below has no device, so every
    miss fails before it could insert; the cache must stay empty and switch off/on cleanly.  (Hits, `from_cache`,
    what is and is not cached, and LRU eviction are pinned on the GPU, tests/test_gpu_parity.py, against the
    reference's own second answers in tests/golden/cache1.json.)"""
    import tempfile
    d = tempfile.mkdtemp(prefix="ns_cache_")
    idx = os.path.join(d, "i")
    nsbind.gen_index(idx, 1, 200, 64, 3, False)
    eng = nsbind.Engine(idx, -1)
    try:
        assert eng.cache_size() == 0
        with pytest.raises(RuntimeError):                     # no device: the search fails and nothing is cached
            eng.search_json("covid", 5)
        assert eng.cache_size() == 0
        eng.set_cache(False)
        eng.set_cache(True)
        assert eng.cache_size() == 0
    finally:
        eng.close()


def test_engine_entries_are_serialised_like_the_reference():
    """The reference takes Engine::mtx in every entry (src/api_engine.cpp:54,:168,:372) because its HTTP layer calls
    search() from a thread pool.  Host-only engine: eight threads hammer search (fails: no device, must not corrupt the
    cache or the error string), query preparation and the cache switch at once; results must equal the serial ones."""
    import tempfile
    import threading
    d = tempfile.mkdtemp(prefix="ns_mt_")
    idx = os.path.join(d, "i")
    nsbind.gen_index(idx, 2, 400, 128, 5, False)
    eng = nsbind.Engine(idx, -1)
    try:
        queries = ["covid virus", "vaccine t000020", "zzzz", "the of", "patients covid covid"] * 40
        serial = eng.build_refs(queries)
        errors = []

        def worker(i):
            try:
                for it in range(30):
                    if i % 3 == 0:
                        try:
                            eng.search_json(queries[(i + it) % len(queries)], 10)
                            errors.append("search without a device succeeded")
                        except RuntimeError as ex:
                            if "no device" not in str(ex):
                                errors.append(str(ex))
                    elif i % 3 == 1:
                        qd, refs, usable = eng.build_refs(queries)
                        if qd.tobytes() != serial[0].tobytes() or refs.tobytes() != serial[1].tobytes() or usable.tobytes() != serial[2].tobytes():
                            errors.append("build_refs differs under concurrency")
                    else:
                        eng.set_cache(it % 2 == 0)
                        eng.cache_size()
            except Exception as ex:   # noqa: BLE001
                errors.append(repr(ex))

        th = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors[:3]
        assert eng.cache_size() == 0
    finally:
        eng.close()


def test_failed_reload_keeps_the_loaded_index_and_corrupt_files_fail_cleanly():
    """The reference swaps its segments in only after every one loaded (src/api_engine.cpp:76-90).  A reload that fails
    half way must leave the engine as it was; corrupt counts in docs.bin / manifest.bin / barrels.bin must end in an
    error return, not in an exception or an allocation of the claimed size crossing the C boundary."""
    import shutil
    import struct
    import tempfile
    d = tempfile.mkdtemp(prefix="ns_reload_")
    idx = os.path.join(d, "i")
    nsbind.gen_index(idx, 2, 300, 64, 9, False)
    eng = nsbind.Engine(idx, -1)
    try:
        before = eng.build_refs(["covid virus", "vaccine"])
        info = eng.segment_info(1)
        seg2 = os.path.join(idx, "segments", "seg_000002")
        shutil.move(os.path.join(seg2, "stats.bin"), os.path.join(d, "stats.keep"))
        with pytest.raises(RuntimeError, match="failed to load segment"):
            eng.reload()
        assert eng.num_segments == 2 and eng.segment_info(1) == info
        after = eng.build_refs(["covid virus", "vaccine"])
        assert all(a.tobytes() == b.tobytes() for a, b in zip(before, after))
        shutil.move(os.path.join(d, "stats.keep"), os.path.join(seg2, "stats.bin"))
        # docs.bin claiming 2^32 - 1 documents
        docs = os.path.join(seg2, "docs.bin")
        raw = open(docs, "rb").read()
        open(docs, "wb").write(struct.pack("<I", 0xFFFFFFFF) + raw[4:])
        with pytest.raises(RuntimeError, match="failed to load segment"):
            eng.reload()
        open(docs, "wb").write(raw)
        # barrels.bin claiming 2^31 barrels
        bpath = os.path.join(seg2, "barrels.bin")
        braw = open(bpath, "rb").read()
        open(bpath, "wb").write(struct.pack("<II", 1 << 31, 1))
        with pytest.raises(RuntimeError, match="failed to load segment"):
            eng.reload()
        open(bpath, "wb").write(braw)
        eng.reload()                                             # everything restored: loads again
        assert eng.num_segments == 2 and eng.segment_info(1) == info
        # a manifest with a corrupt count: no usable names -> falls back to scanning segments/ (src/api_engine.cpp:57-73)
        mpath = os.path.join(idx, "manifest.bin")
        mraw = open(mpath, "rb").read()
        open(mpath, "wb").write(struct.pack("<I", 0xFFFFFFF0) + mraw[4:])
        eng.reload()
        assert eng.num_segments == 2
        # raw postings are read on request only and equal the files
        p0 = eng.segment_postings(0)
        files = sorted(f for f in os.listdir(os.path.join(idx, "segments", "seg_000001")) if f.startswith("inverted_b"))
        cat = b"".join(open(os.path.join(idx, "segments", "seg_000001", f), "rb").read() for f in files)
        assert p0.tobytes() == cat
    finally:
        eng.close()
        shutil.rmtree(d, ignore_errors=True)


def test_tools_and_entry_points_compile():
    """bench.py, __graft_entry__.py and every tool parse (they only run on the GPU box)."""
    import glob
    import py_compile
    for f in [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(ROOT, "tools", "**", "*.py"), recursive=True)):
        py_compile.compile(f, doraise=True)

"""Index inversion (src/lexicon.cpp): the numpy oracle against the REAL reference tool's golden
hashes (CPU), and the device path (ns_invert_forward behind nsbind.invert_segment, the `lexicon <SEGMENT_DIR>`
replacement) against the oracle, file for file (GPU)."""
import base64
import ctypes as C
import hashlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

import forward_gen
import nsbind

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import invert_oracle  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "invert1.json")
REF_TOOL = os.path.join(ROOT, "oracle", "_ref", "lexicon")


def _golden_inputs(seg):
    with open(GOLDEN) as f:
        g = json.load(f)
    os.makedirs(seg, exist_ok=True)
    for name, b64 in g["inputs_base64"].items():
        with open(os.path.join(seg, name), "wb") as f:
            f.write(base64.b64decode(b64))
    return g


def _assert_golden(seg, g, who):
    for name, want in g["outputs"].items():
        b = open(os.path.join(seg, name), "rb").read()
        assert len(b) == want["bytes"], (who, name)
        assert hashlib.sha256(b).hexdigest() == want["sha256"], (who, name)


def _assert_same_files(a, b, who):
    for name in invert_oracle.output_files():
        assert open(os.path.join(a, name), "rb").read() == open(os.path.join(b, name), "rb").read(), (who, name)


def test_oracle_equals_reference_tool_golden(tmp_path):
    seg = str(tmp_path / "seg")
    g = _golden_inputs(seg)
    pairs, kept = invert_oracle.lexicon_tool(seg)
    assert pairs == g["pairs"] and 0 < kept < pairs      # the fixture holds termIds the tool must drop
    _assert_golden(seg, g, "oracle")


def test_generator_is_deterministic(tmp_path):
    """The golden's inputs are what forward_gen writes for the recorded call (the GPU-box tests regenerate
    larger inputs from seeds: the generator must not drift)."""
    seg = str(tmp_path / "seg")
    g = _golden_inputs(str(tmp_path / "g"))
    assert forward_gen.write_inputs(seg, 400, 900, 18, 20261) == g["pairs"]
    for name in ("terms.bin", "forward.bin"):
        assert open(os.path.join(seg, name), "rb").read() == base64.b64decode(g["inputs_base64"][name])


@pytest.mark.skipif(not os.path.exists(REF_TOOL), reason="oracle/_ref/lexicon is built only where /root/reference is mounted")
def test_oracle_equals_reference_tool_on_a_larger_input(tmp_path):
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    forward_gen.write_inputs(a, 20_000, 70_000, 45, 99)
    shutil.copytree(a, b)
    invert_oracle.lexicon_tool(a)
    subprocess.run([REF_TOOL, b], check=True, stderr=subprocess.DEVNULL)
    _assert_same_files(a, b, "oracle vs reference tool")


def test_edge_shapes_oracle(tmp_path):
    """No documents / no terms / no pairs: all 64 + 64 + 1 files exist and are well-formed."""
    for n_docs, n_terms in ((0, 5), (7, 1), (3, 64), (3, 65)):
        seg = str(tmp_path / f"s{n_docs}_{n_terms}")
        forward_gen.write_inputs(seg, n_docs, n_terms, 0 if n_docs == 7 else 3, 1, bad_ids=False, empty_docs=False)
        invert_oracle.lexicon_tool(seg)
        for name in invert_oracle.output_files():
            assert os.path.exists(os.path.join(seg, name))


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_device_inversion_equals_reference_golden(tmp_path):
    seg = str(tmp_path / "seg")
    g = _golden_inputs(seg)
    st = nsbind.invert_segment(seg)
    assert st["pairs"] == g["pairs"] and 0 < st["kept"] < st["pairs"]
    _assert_golden(seg, g, "device")


@pytest.mark.gpu
@pytest.mark.parametrize("n_docs,n_terms,mean,seed", [(20_000, 70_000, 45, 99),      # 17-bit ids: three radix passes
                                                      (5_000, 200, 30, 3),           # one pass, long lists
                                                      (50_000, 65_536, 60, 11),      # exactly 2^16 terms: the drop key needs a 17th bit
                                                      (1, 10, 5, 5), (4097, 3, 2, 8)])
def test_device_inversion_equals_oracle(tmp_path, n_docs, n_terms, mean, seed):
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    forward_gen.write_inputs(a, n_docs, n_terms, mean, seed)
    shutil.copytree(a, b)
    pairs, kept = invert_oracle.lexicon_tool(a)
    st = nsbind.invert_segment(b)
    assert (st["pairs"], st["kept"]) == (pairs, kept)
    _assert_same_files(a, b, f"device vs oracle {n_docs}x{n_terms}")


@pytest.mark.gpu
def test_device_inversion_edges_and_errors(tmp_path):
    for n_docs, n_terms in ((0, 5), (7, 1), (3, 64), (3, 65)):
        a, b = str(tmp_path / f"a{n_docs}_{n_terms}"), str(tmp_path / f"b{n_docs}_{n_terms}")
        forward_gen.write_inputs(a, n_docs, n_terms, 0 if n_docs == 7 else 3, 1, bad_ids=False, empty_docs=False)
        shutil.copytree(a, b)
        invert_oracle.lexicon_tool(a)
        nsbind.invert_segment(b)
        _assert_same_files(a, b, f"edge {n_docs}x{n_terms}")
    with pytest.raises(RuntimeError, match="Missing forward.bin or terms.bin"):
        nsbind.invert_segment(str(tmp_path / "nowhere"))
    # raw ABI: counts that do not add up to n_pairs are refused; duplicate (term, doc) pairs keep file order
    L = nsbind.hip_lib()
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    counts = np.array([2, 0, 3], dtype=np.uint32)
    pairs = np.array([[1, 7], [1, 9], [0, 1], [1, 2], [5, 4]], dtype=np.uint32)
    df = np.zeros(3, dtype=np.uint32)
    out = np.zeros((5, 2), dtype=np.uint32)
    kept = C.c_uint64()
    assert L.ns_invert_forward(ctx, counts.ctypes.data, 3, pairs.ctypes.data, 4, 3, df.ctypes.data, out.ctypes.data, C.byref(kept), None) != 0
    assert L.ns_invert_forward(ctx, counts.ctypes.data, 3, pairs.ctypes.data, 5, 3, df.ctypes.data, out.ctypes.data, C.byref(kept), None) == 0
    assert kept.value == 4 and df.tolist() == [1, 3, 0]
    assert out[:4].tolist() == [[2, 1], [0, 7], [0, 9], [2, 2]]
    L.ns_ctx_destroy(ctx)


@pytest.mark.gpu
def test_inverted_segment_serves_the_hot_path(tmp_path):
    """forward.bin -> device inversion -> the files Engine::reload reads -> scoring: the inverted lists are
    what the scoring kernels traverse (each list's df and docId order checked through the engine)."""
    seg = tmp_path / "index" / "segments" / "seg_000000"
    forward_gen.write_inputs(str(seg), 3000, 500, 25, 17, bad_ids=False)
    nsbind.invert_segment(str(seg))
    counts, pairs = invert_oracle.read_forward(str(seg / "forward.bin"))
    df, postings = invert_oracle.invert(counts, pairs, 500)
    starts = np.concatenate([[0], np.cumsum(df, dtype=np.int64)]).astype(np.int64)
    got = np.concatenate([np.fromfile(str(seg / ("inverted_b%03u.bin" % b)), dtype="<u4") for b in range(64)]).reshape(-1, 2)
    assert np.array_equal(got, postings)
    for t in np.nonzero(df)[0][:50]:
        lst = got[starts[t]:starts[t + 1]]
        assert np.all(np.diff(lst[:, 0].astype(np.int64)) > 0)


@pytest.mark.gpu
def test_inverted_lists_stay_on_the_device_as_a_segment(tmp_path):
    """ns_segment_upload_inverted: the inversion's result becomes a segment's posting stream without leaving the device.
    The segment must serve exactly what a segment uploaded from the oracle's inverted lists serves (hits, nhits, found:
    bytes), its df_out must be the oracle's, dropped termIds shorten the stream, and the state machine refuses misuse."""
    L = nsbind.hip_lib()
    seg = tmp_path / "seg"
    forward_gen.write_inputs(str(seg), 6000, 700, 30, 23, bad_ids=True)      # some termIds >= n_terms: dropped
    counts, pairs = invert_oracle.read_forward(str(seg / "forward.bin"))
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
    n_docs, n_pairs, n_terms = len(counts), len(pairs), 700
    df, postings = invert_oracle.invert(counts, pairs, n_terms)
    postings = np.ascontiguousarray(postings, dtype=np.uint32)
    assert 0 < len(postings) < n_pairs
    doc_len = np.maximum(1, np.bincount(np.repeat(np.arange(n_docs), counts), weights=pairs[:, 1], minlength=n_docs)).astype(np.uint32)
    avgdl = float(np.float32(doc_len.astype(np.float64).mean()))
    ctx = C.c_void_p()
    assert L.ns_ctx_create(0, C.byref(ctx)) == 0
    try:
        # reference segment: the oracle's lists through the ordinary upload
        s_ref = C.c_void_p()
        assert L.ns_segment_upload(ctx, 0, n_docs, C.c_float(avgdl), doc_len.ctypes.data, postings.ctypes.data, postings.nbytes, C.byref(s_ref)) == 0
        # device hand-over
        s_dev = C.c_void_p()
        assert L.ns_segment_upload_begin(ctx, 1, n_docs, C.c_float(avgdl), doc_len.ctypes.data, n_pairs * 8, C.byref(s_dev)) == 0
        df_out = np.zeros(n_terms, dtype=np.uint32)
        host_copy = np.zeros((n_pairs, 2), dtype=np.uint32)
        kept = C.c_uint64()
        assert L.ns_segment_upload_inverted(ctx, s_dev, counts.ctypes.data, pairs.ctypes.data, n_pairs + 1, n_terms, df_out.ctypes.data, None, C.byref(kept), None) != 0
        assert L.ns_segment_upload_inverted(ctx, s_dev, counts.ctypes.data, pairs.ctypes.data, n_pairs, n_terms, df_out.ctypes.data,
                                            host_copy.ctypes.data, C.byref(kept), None) == 0, L.ns_last_error(ctx)
        assert kept.value == len(postings) and np.array_equal(df_out, df.astype(np.uint32)) and np.array_equal(host_copy[:kept.value], postings)
        assert L.ns_segment_upload_inverted(ctx, s_dev, counts.ctypes.data, pairs.ctypes.data, n_pairs, n_terms, df_out.ctypes.data, None, C.byref(kept), None) != 0   # already filled
        assert L.ns_segment_upload_end(ctx, s_dev) == 0, L.ns_last_error(ctx)
        assert L.ns_segment_upload_inverted(ctx, s_dev, counts.ctypes.data, pairs.ctypes.data, n_pairs, n_terms, df_out.ctypes.data, None, C.byref(kept), None) != 0   # published
        starts = np.concatenate([[0], np.cumsum(df, dtype=np.int64)])
        order = np.argsort(-df.astype(np.int64), kind="stable")
        rng = np.random.default_rng(3)
        queries = [[int(order[0]), int(order[1])], [int(order[2])], [int(order[0]), int(order[5]), int(order[300])]]
        queries += [[int(t) for t in rng.choice(np.nonzero(df)[0], size=rng.integers(1, 6), replace=False)] for _ in range(60)]
        res = []
        for sid in (0, 1):
            qd = np.zeros(len(queries), dtype=nsbind.QDESC_DTYPE)
            refs = []
            for qi, q in enumerate(queries):
                qd[qi] = (len(refs), len(q))
                for t in q:
                    refs.append((sid, int(df[t]), int(starts[t]) * 8, 1.0 + (t % 7) * 0.25, 1.0))
            refs = np.array(refs, dtype=nsbind.TERM_DTYPE)
            for k in (10, 100):
                rc, hits, nhits, found = nsbind.search_batch_raw(ctx, qd, refs, k)
                assert rc == 0, L.ns_last_error(ctx)
                res.append((hits["score"].tobytes(), hits["doc"].tobytes(), nhits.tobytes(), found.tobytes()))
        assert res[0] == res[2] and res[1] == res[3]
        assert int(np.frombuffer(res[0][3], dtype=np.uint64).sum()) > 0
    finally:
        L.ns_ctx_destroy(ctx)

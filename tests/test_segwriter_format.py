"""Format parity with the reference's own WRITER (SURVEY 4 tier T0): tests/golden/segwriter1.json holds a segment
that SegmentWriter::write_segment (include/segment_writer.hpp) wrote for a committed logical input — its small files
verbatim, every file's sha256 — and the reference engine's answers over it.
  CPU: the inversion oracle over the writer's forward.bin / terms.bin reproduces the writer's lexicon and inverted
       files; this repo's loader reads the writer's segment (N, avgdl, doc lengths, every list's df / postings).
  GPU: the device inversion reproduces the same files; the engine's answers equal the reference engine's."""
import base64
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import nsbind

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import invert_oracle  # noqa: E402


def _load():
    with open(os.path.join(ROOT, "tests", "golden", "segwriter1.json")) as f:
        return json.load(f)


def _write_inputs(g, index_dir, names):
    seg = os.path.join(index_dir, "segments", "seg_000000")
    os.makedirs(seg, exist_ok=True)
    for n in names:
        with open(os.path.join(seg, n), "wb") as f:
            f.write(base64.b64decode(g["inputs_base64"][n]))
    with open(os.path.join(index_dir, "manifest.bin"), "wb") as f:
        f.write(b"\x01\x00\x00\x00" + b"\x0a\x00\x00\x00" + b"seg_000000")
    return seg


def _assert_files(g, seg, who):
    for name, want in g["files"].items():
        b = open(os.path.join(seg, name), "rb").read()
        assert len(b) == want["bytes"] and hashlib.sha256(b).hexdigest() == want["sha256"], (who, name)


def _spec_docs(g):
    docs = []
    for line in g["spec"].splitlines():
        uid, title, rel, dl, rest = line.split("\t")
        docs.append((uid, int(dl), {t.rsplit(":", 1)[0]: int(t.rsplit(":", 1)[1]) for t in rest.split()}))
    return docs


def test_inversion_oracle_reproduces_the_reference_writer(tmp_path):
    g = _load()
    seg = _write_inputs(g, str(tmp_path / "i"), ("stats.bin", "docs.bin", "forward.bin", "terms.bin"))
    invert_oracle.lexicon_tool(seg)
    _assert_files(g, seg, "oracle")


def test_loader_reads_the_reference_writers_segment(tmp_path):
    g = _load()
    idx = str(tmp_path / "i")
    seg = _write_inputs(g, idx, ("stats.bin", "docs.bin", "forward.bin", "terms.bin"))
    invert_oracle.lexicon_tool(seg)          # byte-identical to the writer's files (previous test)
    docs = _spec_docs(g)
    eng = nsbind.Engine(idx, -1)
    try:
        assert eng.num_segments == 1
        info = eng.segment_info(0)
        assert info["n_docs"] == len(docs)
        total = sum(d[1] for d in docs)
        assert np.float32(info["avgdl"]) == np.float32(np.float32(total) / np.float32(len(docs)))   # :68
        assert eng.segment_doc_len(0).tolist() == [d[1] for d in docs]
        postings = eng.segment_postings(0)
        for term in ("covid", "virus", "t000020", "t000077"):
            want = [(i, d[2][term]) for i, d in enumerate(docs) if term in d[2]]
            e = eng.lookup(0, term)
            if not want:
                assert e is None
                continue
            assert e["df"] == e["count"] == len(want)
            lst = postings[e["byte_off"] // 8: e["byte_off"] // 8 + e["count"]]
            assert [(int(a), int(b)) for a, b in lst] == want
    finally:
        eng.close()


@pytest.mark.gpu
def test_device_inversion_and_search_over_the_reference_writers_segment(tmp_path):
    g = _load()
    idx = str(tmp_path / "i")
    seg = _write_inputs(g, idx, ("stats.bin", "docs.bin", "forward.bin", "terms.bin"))
    nsbind.invert_segment(seg)
    _assert_files(g, seg, "device")
    eng = nsbind.Engine(idx, 0)
    try:
        for case in g["cases"]:
            gh, gn, gf, gu = eng.search_batch(g["queries"], case["k"])
            for qi, ref in enumerate(case["results"]):
                assert int(gf[qi]) == max(ref["found"], 0)
                assert [int(b) for b in gh[qi, : gn[qi]]["score"].view(np.uint32)] == [h[2] for h in ref["hits"]]
    finally:
        eng.close()

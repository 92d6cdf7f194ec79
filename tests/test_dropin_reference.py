"""The drop-in claim, executed: the REAL reference engine with INTEGRATION.md's patch applied.

oracle/_ref/ref_driver_gpu is cord19::Engine compiled from the reference's own sources where they lie, except
src/api_engine.cpp + include/api_engine.hpp, which oracle/patch_reference.py patches in a scratch directory exactly as
INTEGRATION.md section 1 describes (reload() uploads the segments through the C-ABI, search() replaces its scoring section
:426-504 by ns_search_batch), linked against libnextsearch_hip.so (`make -C oracle ref_gpu`, build container only).
Everything around the cut — tokenising, stop-words, the lexicon probes, bm25_idf, the JSON assembly — is the
reference's own code.  Here its answers are compared with the committed goldens, which the UNPATCHED reference wrote:
found and the score bits rank by rank exactly, the docs through the tie-aware comparator (the unpatched
reference's order inside equal-score runs is a hash-table artefact; the patched one returns the C-ABI's canonical order,
which is additionally compared with the oracle's list entry for entry)."""
import os
import subprocess

import numpy as np
import pytest
from conftest import ROOT

import orc

pytestmark = pytest.mark.gpu
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver_gpu")


def _run(index_dir, queries, k, tmp_path):
    qf, of = tmp_path / "q.txt", tmp_path / "out.txt"
    qf.write_text("\n".join(queries) + "\n")
    subprocess.run([DRIVER, "search", index_dir, str(qf), str(k), str(of)], check=True, timeout=300,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    res, lines, i = [], of.read_text().splitlines(), 0
    while i < len(lines):
        tag, found, n = lines[i].split()
        assert tag == "Q"
        hits = []
        for ln in lines[i + 1: i + 1 + int(n)]:
            seg, doc, bits = ln.split()
            hits.append((int(seg), int(doc), int(bits, 16)))
        res.append((int(found), hits))
        i += 1 + int(n)
    return res


@pytest.mark.parametrize("name", ["small2", "mid1", "multi8", "legacy1"])
def test_patched_reference_answers_like_the_unpatched_one(name, golden_index, tmp_path):
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/ref_driver_gpu is not built (make -C oracle ref_gpu needs /root/reference: build container only)")
    g, d, _ = golden_index(name)
    # a query line must survive the driver's line-based input file: the goldens' queries with control bytes are left out
    keep = [i for i, q in enumerate(g["queries"]) if "\n" not in q and "\r" not in q and "\x00" not in q]
    queries = [g["queries"][i] for i in keep]
    ora = orc.Oracle(d)
    try:
        for case in g["cases"]:
            k = case["k"]
            got = _run(d, queries, k, tmp_path)
            assert len(got) == len(queries)
            oh, on, of, ou = ora.search_batch(queries, k)
            for j, qi in enumerate(keep):
                ref = case["results"][qi]
                found, hits = got[j]
                assert found == ref["found"], (name, k, queries[j])
                if ref["found"] < 0:
                    assert hits == []
                    continue
                ref_hits = [tuple(h) for h in ref["hits"]]
                assert [b for _, _, b in hits] == [b for _, _, b in ref_hits], (name, k, queries[j])
                ok, why = orc.tie_aware_equal(hits, found, ora, queries[j], k)
                assert ok, (name, k, queries[j], why)
                mine = [(int(h["seg"]), int(h["doc"]), int(orc.f32_bits(h["score"]))) for h in oh[j, : on[j]]]
                assert hits == mine, (name, k, queries[j])
    finally:
        ora.close()

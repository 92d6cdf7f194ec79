"""Deterministic forward.bin / terms.bin inputs for the inversion step (src/lexicon.cpp's inputs), in the
reference's format: terms.bin = u32 n + n x (u32 len, bytes); forward.bin = u32 numDocs + per doc
u32 cnt + cnt x (u32 termId, u32 tf).  Used by tests, tools/gen_golden.py and tools/invert_bench.py."""
import os
import struct

import numpy as np


def write_inputs(seg_dir, n_docs, n_terms, mean_terms, seed, bad_ids=True, empty_docs=True):
    """Zipf-ish term choice (unique per document), geometric tf; some documents empty, some termIds
    >= n_terms (the reference drops them, src/lexicon.cpp:69), many terms never used."""
    rng = np.random.default_rng(seed)
    os.makedirs(seg_dir, exist_ok=True)
    terms = [("t%06d" % i).encode() if i % 7 else ("tërm-%d" % i).encode("utf-8") for i in range(n_terms)]
    with open(os.path.join(seg_dir, "terms.bin"), "wb") as f:
        f.write(struct.pack("<I", n_terms) + b"".join(struct.pack("<I", len(t)) + t for t in terms))
    counts = rng.poisson(mean_terms, size=n_docs).astype(np.uint32)
    if n_docs == 0 or int(counts.sum()) == 0:
        np.concatenate([[n_docs], np.zeros(n_docs)]).astype("<u4").tofile(os.path.join(seg_dir, "forward.bin"))
        return 0
    if empty_docs and n_docs > 10:
        counts[rng.choice(n_docs, size=max(1, n_docs // 50), replace=False)] = 0
        counts[0] = 0
        counts[-1] = 0
    total = int(counts.sum())
    # rank ~ 1/r over [0, 0.9 * n_terms): the tail of the dictionary stays unused
    u = rng.random(total)
    ids = np.minimum((np.exp(u * np.log(max(2.0, 0.9 * n_terms))) - 1.0).astype(np.uint32), np.uint32(n_terms - 1))
    doc = np.repeat(np.arange(n_docs, dtype=np.uint32), counts)
    # unique termIds per document: drop repeats (keeps file order)
    key = doc.astype(np.uint64) << np.uint64(32) | ids.astype(np.uint64)
    _, first = np.unique(key, return_index=True)
    keep = np.zeros(total, dtype=bool)
    keep[first] = True
    ids, doc = ids[keep], doc[keep]
    tf = rng.geometric(0.45, size=len(ids)).astype(np.uint32)
    if bad_ids and len(ids) > 100:
        bad = rng.choice(len(ids), size=max(1, len(ids) // 200), replace=False)
        ids[bad] = n_terms + (rng.integers(0, 5, size=len(bad)).astype(np.uint32) * np.uint32(1000003))
        ids[bad[0]] = 0xFFFFFFFF
    counts = np.bincount(doc, minlength=n_docs).astype(np.uint32)
    out = np.empty(1 + n_docs + 2 * len(ids), dtype="<u4")
    out[0] = n_docs
    starts = 1 + np.arange(n_docs, dtype=np.int64) + 2 * np.concatenate([[0], np.cumsum(counts[:-1], dtype=np.int64)])
    out[starts] = counts
    pair_pos = np.repeat(starts + 1, counts) + 2 * (np.arange(len(ids), dtype=np.int64) - np.repeat(np.concatenate([[0], np.cumsum(counts[:-1], dtype=np.int64)]), counts))
    out[pair_pos] = ids
    out[pair_pos + 1] = tf
    out.tofile(os.path.join(seg_dir, "forward.bin"))
    return len(ids)

"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of the reference's `lexicon <SEGMENT_DIR>` tool,
src/lexicon.cpp: terms.bin + forward.bin -> barrels.bin, lexicon_bNNN.bin, inverted_bNNN.bin.

Only tests/ and tools/invert_bench.py's cpu_baseline leg may import this; the product path
(nextsearch-api_amd/csrc/ns_invert.hip behind ns_invert_forward) never does.

PINNED: tests/golden/invert1.json holds the sha256 of every file the REAL tool (oracle/_ref/lexicon,
g++ on /root/reference/src/lexicon.cpp as it lies) wrote for the committed fixture; tests/test_invert_cpu.py
checks this restatement against them, and against the real tool on a larger seeded input whenever
oracle/_ref/lexicon is present.

What is restated (file:line of the reference):
  :44-48   terms.bin  = u32 n, n x (u32 len, bytes)
  :60-72   forward.bin = u32 numDocs, per doc u32 cnt, cnt x (u32 termId, u32 tf); pairs with
           termId >= n are skipped (:69); docIds are the running document index
  :86-90   64 barrels, terms_per_barrel = ceil(n / 64), at least 1; barrels.bin = (u32 64, u32 tpb)
  :107-127 per termId in ascending order, skipping empty lists: std::sort by docId (UNSTABLE: the order
           of equal docIds is unspecified in the reference; here and on the device it is file order),
           lexicon record (string term, u32 termId, u32 df, u64 byte offset in the barrel's inverted
           file, u32 df), then the postings (u32 docId, u32 tf)
  :131-146 each lexicon file starts with the u32 number of records it holds (all 64 files exist)
"""
import os
import struct

import numpy as np

BARREL_COUNT = 64


def read_terms(path):
    with open(path, "rb") as f:
        b = f.read()
    (n,) = struct.unpack_from("<I", b, 0)
    pos, terms = 4, []
    for _ in range(n):
        (ln,) = struct.unpack_from("<I", b, pos)
        terms.append(b[pos + 4: pos + 4 + ln])
        pos += 4 + ln
    return terms


def read_forward(path):
    """-> (counts u32[numDocs], pairs u32[total, 2])"""
    a = np.fromfile(path, dtype="<u4")
    n_docs = int(a[0])
    counts = np.empty(n_docs, dtype=np.uint32)
    chunks, pos = [], 1
    for d in range(n_docs):
        c = int(a[pos])
        counts[d] = c
        chunks.append(a[pos + 1: pos + 1 + 2 * c])
        pos += 1 + 2 * c
    flat = np.concatenate(chunks) if chunks else np.empty(0, dtype=np.uint32)
    return counts, flat.reshape(-1, 2)


def invert(counts, pairs, n_terms):
    """-> (df u32[n_terms], postings u32[kept, 2] = lists in termId order, each in docId (file) order)"""
    doc = np.repeat(np.arange(len(counts), dtype=np.uint32), counts)
    keep = pairs[:, 0] < n_terms
    t, d, tf = pairs[keep, 0], doc[keep], pairs[keep, 1]
    order = np.argsort(t, kind="stable")          # documents are already in docId order
    df = np.bincount(t, minlength=n_terms).astype(np.uint32)
    return df, np.stack([d[order], tf[order]], axis=1).astype("<u4")


def write_segment(seg_dir, terms, df, postings):
    n = len(terms)
    tpb = (n + BARREL_COUNT - 1) // BARREL_COUNT or 1
    with open(os.path.join(seg_dir, "barrels.bin"), "wb") as f:
        f.write(struct.pack("<II", BARREL_COUNT, tpb))
    starts = np.concatenate([[0], np.cumsum(df.astype(np.uint64))]).astype(np.uint64)
    for b in range(BARREL_COUNT):
        lo = min(b * tpb, n)
        hi = n if b == BARREL_COUNT - 1 else min((b + 1) * tpb, n)
        recs, off, cnt = [], 0, 0
        for tid in range(lo, hi):
            c = int(df[tid])
            if c == 0:
                continue
            cnt += 1
            recs.append(struct.pack("<I", len(terms[tid])) + terms[tid] + struct.pack("<IIQI", tid, c, off, c))
            off += c * 8
        with open(os.path.join(seg_dir, "lexicon_b%03u.bin" % b), "wb") as f:
            f.write(struct.pack("<I", cnt) + b"".join(recs))
        with open(os.path.join(seg_dir, "inverted_b%03u.bin" % b), "wb") as f:
            f.write(postings[int(starts[lo]): int(starts[hi])].tobytes())


def lexicon_tool(seg_dir):
    """What `lexicon <SEGMENT_DIR>` does."""
    terms = read_terms(os.path.join(seg_dir, "terms.bin"))
    counts, pairs = read_forward(os.path.join(seg_dir, "forward.bin"))
    df, postings = invert(counts, pairs, len(terms))
    write_segment(seg_dir, terms, df, postings)
    return len(pairs), len(postings)


def output_files():
    return ["barrels.bin"] + ["lexicon_b%03u.bin" % b for b in range(BARREL_COUNT)] + ["inverted_b%03u.bin" % b for b in range(BARREL_COUNT)]

"""TEST INFRASTRUCTURE: ctypes binding of the CPU oracle (oracle/libbm25_oracle.so) plus the
tie-aware comparator of SURVEY.md §8(c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libbm25_oracle.so")
REF_DRIVER = os.path.join(ORACLE_DIR, "_ref", "ref_driver")

HIT_DTYPE = np.dtype([("score", "<f4"), ("seg", "<u4"), ("doc", "<u4")])
FLAG_OR, FLAG_AND = 0, 1

_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, u32 = C.c_void_p, C.c_uint32
        L.orc_open.argtypes = [C.c_char_p]
        L.orc_open.restype = vp
        L.orc_close.argtypes = [vp]
        L.orc_close.restype = None
        L.orc_error.restype = C.c_char_p
        L.orc_num_segments.argtypes = [vp]
        L.orc_num_segments.restype = u32
        L.orc_segment_docs.argtypes = [vp, u32]
        L.orc_segment_docs.restype = u32
        L.orc_search.argtypes = [vp, C.c_char_p, C.c_int, u32, vp, C.POINTER(u32), C.POINTER(C.c_uint64)]
        L.orc_search_batch.argtypes = [vp, C.POINTER(C.c_char_p), u32, C.c_int, u32, vp, vp, vp, vp, C.c_int]
        L.orc_scores.argtypes = [vp, C.c_char_p, u32, u32, vp, vp]
        L.orc_query_postings.argtypes = [vp, C.c_char_p]
        L.orc_query_postings.restype = C.c_uint64
        _lib = L
    return _lib


def clamp_k(k):
    return max(1, min(int(k), 100))


class Oracle:
    def __init__(self, index_dir):
        self.L = lib()
        self.h = self.L.orc_open(index_dir.encode())
        if not self.h:
            raise RuntimeError("orc_open: " + self.L.orc_error().decode())

    def close(self):
        if self.h:
            self.L.orc_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @property
    def num_segments(self):
        return self.L.orc_num_segments(self.h)

    def segment_docs(self, seg):
        return self.L.orc_segment_docs(self.h, seg)

    def search_batch(self, queries, k, flags=FLAG_OR, threads=8):
        Q, K = len(queries), clamp_k(k)
        arr = (C.c_char_p * Q)()
        arr[:] = [q.encode() for q in queries]
        hits = np.zeros((Q, K), dtype=HIT_DTYPE)
        hits["score"] = -np.inf
        hits["seg"] = 0xFFFFFFFF
        hits["doc"] = 0xFFFFFFFF
        tmp = np.zeros((Q, K), dtype=HIT_DTYPE)
        nhits = np.zeros(Q, dtype=np.uint32)
        found = np.zeros(Q, dtype=np.uint64)
        usable = np.zeros(Q, dtype=np.uint8)
        rc = self.L.orc_search_batch(self.h, arr, Q, k, flags, tmp.ctypes.data, nhits.ctypes.data, found.ctypes.data,
                                     usable.ctypes.data, threads)
        assert rc == 0
        for q in range(Q):
            hits[q, : nhits[q]] = tmp[q, : nhits[q]]
        return hits, nhits, found, usable

    def scores(self, query, seg, flags=FLAG_OR):
        n = self.segment_docs(seg)
        acc = np.zeros(max(n, 1), dtype=np.float32)
        touched = np.zeros(max(n, 1), dtype=np.uint8)
        rc = self.L.orc_scores(self.h, query.encode(), flags, seg, acc.ctypes.data, touched.ctypes.data)
        assert rc >= 0
        return acc[:n], touched[:n].astype(bool)

    def query_postings(self, query):
        return self.L.orc_query_postings(self.h, query.encode())


def f32_bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def parse_driver_output(path):
    """Parse the 'Q <found|-1> <n>' / '<seg> <doc> <bits-hex>' format of ref_driver / bm25_oracle_cli."""
    out = []
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i].strip()
        i += 1
        if not ln:
            continue
        tag, found, n = ln.split()
        assert tag == "Q"
        hits = []
        for _ in range(int(n)):
            s, d, b = lines[i].split()
            i += 1
            hits.append((int(s), int(d), int(b, 16)))
        out.append({"found": int(found), "hits": hits})
    return out


def run_ref_driver(index_dir, queries, k, workdir):
    """Run the REAL reference engine (only where oracle/_ref/ref_driver exists)."""
    qpath = os.path.join(workdir, "queries.txt")
    opath = os.path.join(workdir, "ref_out.txt")
    with open(qpath, "w") as f:
        f.write("\n".join(queries) + "\n")
    subprocess.run([REF_DRIVER, "search", index_dir, qpath, str(k), opath], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    res = parse_driver_output(opath)
    # a trailing empty query line is dropped by the text format; pad
    while len(res) < len(queries):
        res.append({"found": -1, "hits": []})
    return res


def batch_digests(hits, nhits, found, block=1024):
    """Digests of a whole batch's results (shared by tools/gen_golden.py and tests/test_gpu_parity.py).
    exact[b]: SHA-256 over block b's found, nhits and the valid hits' (score bits, seg, doc) in rank order — the
              canonical order (score desc, seg asc, doc asc), what the oracle and the HIP path return;
    ties[b]:  SHA-256 over found, nhits and each query's score bits SORTED — invariant under the order inside
              equal-score runs, which the reference leaves to its hash table: comparable with the REAL
              reference's answers."""
    import hashlib
    Q, K = hits.shape
    out = {"exact": [], "ties": []}
    for b0 in range(0, Q, block):
        he, ht = hashlib.sha256(), hashlib.sha256()
        b1 = min(Q, b0 + block)
        head = np.asarray(found[b0:b1], dtype="<u8").tobytes() + np.asarray(nhits[b0:b1], dtype="<u4").tobytes()
        he.update(head)
        ht.update(head)
        for q in range(b0, b1):
            n = int(nhits[q])
            h = hits[q, :n]
            bits = h["score"].view(np.uint32)
            he.update(np.stack([bits, h["seg"], h["doc"]], axis=1).astype("<u4").tobytes())
            ht.update(np.sort(bits).astype("<u4").tobytes())
        out["exact"].append(he.hexdigest())
        out["ties"].append(ht.hexdigest())
    return out


def tie_aware_equal(got_hits, got_found, oracle, query, k, flags=FLAG_OR):
    """SURVEY.md §8(c) tie policy.  `got_hits` is a list of (seg, doc, score_bits) in rank order from an
    implementation whose order inside equal-score runs is unspecified (the real reference).  It is
    accepted iff: found matches; every returned doc's score bit pattern equals the oracle's fp32 sum
    for that doc; scores are non-increasing; and the multiset of score bit patterns equals the
    oracle's canonical top-K (so a run cut by K holds the right NUMBER of tied docs, and every
    returned doc of it is a genuine member of the tied candidate set)."""
    K = clamp_k(k)
    per_seg = {}
    for s in range(oracle.num_segments):
        per_seg[s] = oracle.scores(query, s, flags)
    total_found = int(sum(int(t.sum()) for _, t in per_seg.values()))
    if got_found != total_found:
        return False, f"found {got_found} != {total_found}"
    allscores = np.concatenate([a[t] for a, t in per_seg.values()]) if per_seg else np.zeros(0, np.float32)
    top = np.sort(allscores)[::-1][:K]
    if len(got_hits) != len(top):
        return False, f"nhits {len(got_hits)} != {len(top)}"
    seen = set()
    prev = None
    for rank, (seg, doc, bits) in enumerate(got_hits):
        acc, touched = per_seg[seg]
        if not touched[doc]:
            return False, f"rank {rank}: doc {seg}:{doc} is not a candidate"
        if int(f32_bits(acc[doc : doc + 1])[0]) != bits:
            return False, f"rank {rank}: score bits {bits:08x} != oracle {int(f32_bits(acc[doc:doc+1])[0]):08x}"
        if (seg, doc) in seen:
            return False, f"rank {rank}: duplicate doc {seg}:{doc}"
        seen.add((seg, doc))
        sc = np.array([bits], dtype=np.uint32).view(np.float32)[0]
        if prev is not None and sc > prev:
            return False, f"rank {rank}: scores not sorted"
        prev = sc
    got_bits = sorted(b for _, _, b in got_hits)
    want_bits = sorted(int(x) for x in f32_bits(top))
    if got_bits != want_bits:
        return False, "score multiset differs from oracle top-K"
    return True, ""

"""ctypes bindings of the two in-tree libraries (plumbing only; no compute happens in Python).

  libnextsearch_hip.so   include/nextsearch_hip.h   C-ABI of the MI355X hot path
  libnextsearch_host.so  include/nextsearch_host.h  C wrappers of the host facade (Engine mirror)

Loading fails loudly if the libraries are missing: there is no Python or CPU fallback.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# NS_HIP_LIB: another build of the HIP library for THIS process (tests/variants: libnextsearch_hip_variants.so; A/B runs).
# It is loaded first and globally, and carries the product library's SONAME, so the host library binds to it too.
HIP_LIB_PATH = os.environ.get("NS_HIP_LIB") or os.path.join(HERE, "libnextsearch_hip.so")
HOST_LIB_PATH = os.path.join(HERE, "libnextsearch_host.so")

NS_OK = 0
NS_FLAG_OR = 0
NS_FLAG_AND = 1
NS_INFO_IMPACTS = 0x100
NS_INFO_PACKED = 0x200
NS_INFO_PRUNED = 0x400
NS_INFO_SHARED = 0x800
NS_MAX_K = 100


class NsTermRef(C.Structure):
    _fields_ = [("seg_id", C.c_uint32), ("count", C.c_uint32), ("byte_off", C.c_uint64),
                ("idf", C.c_float), ("qweight", C.c_float)]


class NsQueryDesc(C.Structure):
    _fields_ = [("term_begin", C.c_uint32), ("term_count", C.c_uint32)]


class NsHit(C.Structure):
    _fields_ = [("score", C.c_float), ("seg_id", C.c_uint32), ("doc_id", C.c_uint32)]


class NsBatchInfo(C.Structure):
    _fields_ = [("postings", C.c_uint64), ("algo_bytes", C.c_uint64), ("n_queries", C.c_uint32),
                ("n_items", C.c_uint32), ("n_term_refs", C.c_uint32), ("tile_docs", C.c_uint32),
                ("k", C.c_uint32), ("flags", C.c_uint32), ("last_score_kernel_ms", C.c_float),
                ("last_total_ms", C.c_float), ("timed_runs", C.c_uint32), ("shared_lists", C.c_uint32),
                ("sum_score_kernel_ms", C.c_double), ("sum_total_ms", C.c_double), ("shared_postings", C.c_uint64)]


HIT_DTYPE = np.dtype([("score", "<f4"), ("seg", "<u4"), ("doc", "<u4")])
TERM_DTYPE = np.dtype([("seg_id", "<u4"), ("count", "<u4"), ("byte_off", "<u8"), ("idf", "<f4"), ("qweight", "<f4")])
QDESC_DTYPE = np.dtype([("term_begin", "<u4"), ("term_count", "<u4")])
assert HIT_DTYPE.itemsize == C.sizeof(NsHit) == 12
assert TERM_DTYPE.itemsize == C.sizeof(NsTermRef) == 24
assert QDESC_DTYPE.itemsize == C.sizeof(NsQueryDesc) == 8

# every symbol include/nextsearch_hip.h declares
HIP_SYMBOLS = [
    "ns_ctx_create", "ns_ctx_destroy", "ns_ctx_set_stream", "ns_last_error", "ns_device_name",
    "ns_segment_upload", "ns_segment_release", "ns_segment_upload_begin", "ns_segment_upload_append", "ns_segment_upload_end", "ns_search_batch", "ns_batch_prepare",
    "ns_batch_bind_outputs", "ns_batch_run", "ns_batch_stream", "ns_batch_gap_ms", "ns_batch_sync", "ns_batch_fetch", "ns_batch_get_info",
    "ns_batch_destroy", "ns_set_tuning", "ns_segment_build_impacts", "ns_ctx_use_impacts", "ns_ctx_set_host_threads", "ns_ctx_set_overlap", "ns_segment_build_packed", "ns_ctx_use_packed", "ns_segment_build_skips", "ns_ctx_use_skips", "ns_segment_build_blockmax", "ns_ctx_use_pruning", "ns_ctx_use_merge", "ns_ctx_share_scores",
    "ns_invert_forward", "ns_segment_upload_inverted", "ns_merge_rank_rows", "ns_sem_upload", "ns_sem_release", "ns_sem_topk",
]
HOST_SYMBOLS = [
    "nsh_gen_index", "nsh_engine_open", "nsh_engine_open_multi", "nsh_engine_num_devices", "nsh_shard_bounds", "nsh_engine_close", "nsh_engine_reload", "nsh_engine_error", "nsh_engine_ctx",
    "nsh_engine_num_segments", "nsh_engine_segment_name", "nsh_engine_segment_info",
    "nsh_engine_segment_doc_len", "nsh_engine_segment_postings", "nsh_engine_lookup", "nsh_bm25_idf",
    "nsh_base_terms", "nsh_engine_build_refs", "nsh_engine_search_json", "nsh_free",
    "nsh_engine_search_batch", "nsh_engine_prepare", "nsh_engine_doc_metadata", "nsh_engine_hits_to_json", "nsh_engine_search_batch_json",
    "nsh_engine_build_impacts", "nsh_engine_use_impacts", "nsh_engine_build_packed", "nsh_engine_use_packed", "nsh_engine_build_blockmax", "nsh_engine_use_pruning", "nsh_engine_use_merge", "nsh_engine_share_scores", "nsh_engine_use_skips", "nsh_invert_segment", "nsh_invert_error",
    "nsh_engine_semantic_info", "nsh_engine_expand", "nsh_engine_semantic_row", "nsh_engine_set_cache", "nsh_engine_cache_size",
]

_hip = None
_host = None


def hip_lib():
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise RuntimeError(f"{HIP_LIB_PATH} is missing: build it with `make -C nextsearch-api_amd` "
                               "(there is no fallback path)")
        L = C.CDLL(HIP_LIB_PATH, mode=C.RTLD_GLOBAL)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.ns_ctx_create.argtypes = [i32, C.POINTER(vp)]
        L.ns_ctx_destroy.argtypes = [vp]
        L.ns_ctx_destroy.restype = None
        L.ns_ctx_set_stream.argtypes = [vp, vp]
        L.ns_last_error.argtypes = [vp]
        L.ns_last_error.restype = C.c_char_p
        L.ns_device_name.argtypes = [vp]
        L.ns_device_name.restype = C.c_char_p
        L.ns_segment_upload.argtypes = [vp, u32, u32, C.c_float, vp, vp, u64, C.POINTER(vp)]
        L.ns_segment_release.argtypes = [vp, vp]
        L.ns_segment_upload_begin.argtypes = [vp, u32, u32, C.c_float, vp, u64, C.POINTER(vp)]
        L.ns_segment_upload_append.argtypes = [vp, vp, vp, u64]
        L.ns_segment_upload_end.argtypes = [vp, vp]
        L.ns_segment_build_impacts.argtypes = [vp, vp, vp, vp, vp, u32]
        L.ns_ctx_use_impacts.argtypes = [vp, i32]
        L.ns_segment_build_skips.argtypes = [vp, vp, vp, vp, u32]
        L.ns_ctx_use_skips.argtypes = [vp, i32]
        L.ns_ctx_set_host_threads.argtypes = [vp, u32]
        L.ns_ctx_set_overlap.argtypes = [vp, i32]
        L.ns_segment_build_packed.argtypes = [vp, vp]
        L.ns_ctx_use_packed.argtypes = [vp, i32]
        L.ns_segment_build_blockmax.argtypes = [vp, vp, vp, vp, vp, u32]
        L.ns_ctx_use_pruning.argtypes = [vp, i32]
        L.ns_ctx_use_merge.argtypes = [vp, i32]
        L.ns_ctx_share_scores.argtypes = [vp, i32]
        L.ns_sem_upload.argtypes = [vp, vp, u32, u32, C.POINTER(vp)]
        L.ns_sem_release.argtypes = [vp, vp]
        L.ns_sem_topk.argtypes = [vp, vp, vp, u32, u32, C.c_float, vp, vp, vp, vp, vp, vp]
        L.ns_merge_rank_rows.argtypes = [vp, vp, vp, vp, u32, u32, u32, vp, u32, vp, vp, vp]
        L.ns_invert_forward.argtypes = [vp, vp, u32, vp, u64, u32, vp, vp, C.POINTER(u64), vp]
        L.ns_segment_upload_inverted.argtypes = [vp, vp, vp, vp, u64, u32, vp, vp, C.POINTER(u64), vp]
        L.ns_search_batch.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp, u32]
        L.ns_batch_prepare.argtypes = [vp, vp, vp, u32, u32, u32, C.POINTER(vp)]
        L.ns_batch_bind_outputs.argtypes = [vp, vp, vp, vp]
        L.ns_batch_run.argtypes = [vp, i32]
        L.ns_batch_gap_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
        L.ns_batch_stream.argtypes = [vp]
        L.ns_batch_stream.restype = vp
        L.ns_batch_sync.argtypes = [vp]
        L.ns_batch_fetch.argtypes = [vp, vp, vp, vp]
        L.ns_batch_get_info.argtypes = [vp, C.POINTER(NsBatchInfo)]
        L.ns_batch_destroy.argtypes = [vp]
        L.ns_batch_destroy.restype = None
        L.ns_set_tuning.argtypes = [vp, u32, u32, u32]
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: build it with `make -C nextsearch-api_amd`")
        L = C.CDLL(HOST_LIB_PATH, mode=C.RTLD_GLOBAL)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.nsh_gen_index.argtypes = [C.c_char_p, u32, u32, u32, u64, i32, C.POINTER(u64)]
        L.nsh_engine_open.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
        L.nsh_engine_open_multi.argtypes = [C.c_char_p, C.POINTER(i32), u32, C.POINTER(vp)]
        L.nsh_engine_num_devices.argtypes = [vp]
        L.nsh_engine_num_devices.restype = u32
        L.nsh_shard_bounds.argtypes = [C.c_uint64, u32, u32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.nsh_shard_bounds.restype = None
        L.nsh_engine_close.argtypes = [vp]
        L.nsh_engine_close.restype = None
        L.nsh_engine_reload.argtypes = [vp]
        L.nsh_engine_error.argtypes = [vp]
        L.nsh_engine_error.restype = C.c_char_p
        L.nsh_engine_ctx.argtypes = [vp]
        L.nsh_engine_ctx.restype = vp
        L.nsh_engine_num_segments.argtypes = [vp]
        L.nsh_engine_num_segments.restype = u32
        L.nsh_engine_segment_name.argtypes = [vp, u32]
        L.nsh_engine_segment_name.restype = C.c_char_p
        L.nsh_engine_segment_info.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(C.c_float), C.POINTER(u64),
                                              C.POINTER(u32), C.POINTER(i32)]
        L.nsh_engine_segment_doc_len.argtypes = [vp, u32]
        L.nsh_engine_segment_doc_len.restype = vp
        L.nsh_engine_segment_postings.argtypes = [vp, u32, C.POINTER(u64)]
        L.nsh_engine_segment_postings.restype = vp
        L.nsh_engine_lookup.argtypes = [vp, u32, C.c_char_p, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32),
                                        C.POINTER(u64), C.POINTER(C.c_float)]
        L.nsh_bm25_idf.argtypes = [u32, u32]
        L.nsh_bm25_idf.restype = C.c_float
        L.nsh_base_terms.argtypes = [C.c_char_p, C.c_char_p, u32]
        L.nsh_base_terms.restype = u32
        L.nsh_engine_build_refs.argtypes = [vp, C.POINTER(C.c_char_p), u32, vp, vp, u32, C.POINTER(u32), vp]
        L.nsh_engine_search_json.argtypes = [vp, C.c_char_p, i32, C.POINTER(vp)]
        L.nsh_free.argtypes = [vp]
        L.nsh_free.restype = None
        L.nsh_engine_search_batch.argtypes = [vp, C.POINTER(C.c_char_p), u32, i32, u32, vp, vp, vp, vp]
        L.nsh_engine_prepare.argtypes = [vp, C.POINTER(C.c_char_p), u32, i32, u32, C.POINTER(vp)]
        L.nsh_engine_doc_metadata.argtypes = [vp, u32, u32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
        L.nsh_engine_hits_to_json.argtypes = [vp, C.c_char_p, i32, i32, u64, vp, u32, C.POINTER(vp)]
        L.nsh_engine_search_batch_json.argtypes = [vp, C.POINTER(C.c_char_p), u32, i32, C.POINTER(vp), vp]
        L.nsh_invert_segment.argtypes = [C.c_char_p, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.nsh_invert_error.restype = C.c_char_p
        L.nsh_engine_set_cache.argtypes = [vp, i32]
        L.nsh_engine_set_cache.restype = None
        L.nsh_engine_cache_size.argtypes = [vp]
        L.nsh_engine_cache_size.restype = u32
        L.nsh_engine_semantic_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
        L.nsh_engine_expand.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
        L.nsh_engine_semantic_row.argtypes = [vp, u32, C.POINTER(C.c_char_p), C.POINTER(vp)]
        L.nsh_engine_build_impacts.argtypes = [vp]
        L.nsh_engine_use_impacts.argtypes = [vp, i32]
        L.nsh_engine_use_impacts.restype = None
        L.nsh_engine_use_skips.argtypes = [vp, i32]
        L.nsh_engine_use_skips.restype = None
        L.nsh_engine_build_packed.argtypes = [vp]
        L.nsh_engine_use_packed.argtypes = [vp, i32]
        L.nsh_engine_use_packed.restype = None
        L.nsh_engine_build_blockmax.argtypes = [vp]
        L.nsh_engine_use_pruning.argtypes = [vp, i32]
        L.nsh_engine_use_pruning.restype = None
        L.nsh_engine_use_merge.argtypes = [vp, i32]
        L.nsh_engine_use_merge.restype = None
        L.nsh_engine_share_scores.argtypes = [vp, i32]
        L.nsh_engine_share_scores.restype = None
        _host = L
    return _host


def _cstr_array(strings):
    arr = (C.c_char_p * len(strings))()
    arr[:] = [s.encode("utf-8") if isinstance(s, str) else s for s in strings]
    return arr


def clamp_k(k):
    return max(1, min(int(k), NS_MAX_K))


def gen_index(index_dir, n_segments, docs_per_segment, vocab=65536, seed=1337, legacy=False):
    total = C.c_uint64(0)
    rc = host_lib().nsh_gen_index(index_dir.encode(), n_segments, docs_per_segment, vocab, seed, int(legacy), C.byref(total))
    if rc != 0:
        raise RuntimeError(f"nsh_gen_index({index_dir}) failed")
    return total.value


class Batch:
    """Staged batch: descriptors resident on the device; run() enqueues one pass of the hot path."""

    def __init__(self, handle, n_queries, k):
        self.h = handle
        self.Q = n_queries
        self.K = k

    def bind_outputs(self, d_hits, d_nhits, d_found):
        rc = hip_lib().ns_batch_bind_outputs(self.h, d_hits, d_nhits, d_found)
        if rc != NS_OK:
            raise RuntimeError("ns_batch_bind_outputs failed")

    def run(self, timed=False, fetch=False):
        """fetch=True (NS_RUN_FETCH): the results' copy to pinned host memory rides behind the kernels and fetch()
        waits for this batch only — batches can then overlap on one ctx."""
        rc = hip_lib().ns_batch_run(self.h, int(bool(timed)) | (2 if fetch else 0))
        if rc != NS_OK:
            raise RuntimeError(f"ns_batch_run failed rc={rc}")

    @property
    def stream(self):
        """hipStream_t (integer) the batch's work goes to."""
        return hip_lib().ns_batch_stream(self.h)

    def sync(self):
        rc = hip_lib().ns_batch_sync(self.h)
        if rc != NS_OK:
            raise RuntimeError(f"ns_batch_sync failed rc={rc}")

    def fetch(self):
        hits = np.empty((self.Q, self.K), dtype=HIT_DTYPE)
        nhits = np.empty(self.Q, dtype=np.uint32)
        found = np.empty(self.Q, dtype=np.uint64)
        rc = hip_lib().ns_batch_fetch(self.h, hits.ctypes.data, nhits.ctypes.data, found.ctypes.data)
        if rc != NS_OK:
            raise RuntimeError(f"ns_batch_fetch failed rc={rc}")
        return hits, nhits, found

    def fetch_into(self, hits, nhits, found):
        """fetch() into caller-owned arrays (HIT_DTYPE [Q, K], uint32 [Q], uint64 [Q]): no allocation per batch."""
        rc = hip_lib().ns_batch_fetch(self.h, hits.ctypes.data, nhits.ctypes.data, found.ctypes.data)
        if rc != NS_OK:
            raise RuntimeError(f"ns_batch_fetch failed rc={rc}")

    def info(self):
        inf = NsBatchInfo()
        hip_lib().ns_batch_get_info(self.h, C.byref(inf))
        return inf

    def close(self):
        if self.h:
            hip_lib().ns_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Engine:
    """Python view of the host facade (mirror of cord19::Engine: reload at open, search, search_batch)."""

    def __init__(self, index_dir, device=0):
        """device: one device id (< 0: host-only), or a list of device ids: the multi-device engine (index replicated,
        batches cut into contiguous shards, one host thread + context per device)."""
        self._L = host_lib()
        h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            arr = (C.c_int32 * len(device))(*device)
            rc = self._L.nsh_engine_open_multi(index_dir.encode(), arr, len(device), C.byref(h))
        else:
            rc = self._L.nsh_engine_open(index_dir.encode(), device, C.byref(h))
        self.h = h
        if rc != 0:
            msg = self._L.nsh_engine_error(h).decode()
            self._L.nsh_engine_close(h)
            self.h = None
            raise RuntimeError(f"Engine.reload failed: {msg}")
        self.device = device

    def close(self):
        if self.h:
            close_batches_of(self.ctx)   # a batch must not outlive its device context
            self._L.nsh_engine_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_devices(self):
        return self._L.nsh_engine_num_devices(self.h)

    def reload(self):
        """Engine::reload() on the same directory; on failure the engine keeps what it had."""
        old_ctx = self.ctx
        if self._L.nsh_engine_reload(self.h) != 0:
            raise RuntimeError(f"Engine.reload failed: {self.error()}")
        if _ctx_key(self.ctx) != _ctx_key(old_ctx):
            _LIVE_BATCHES.pop(_ctx_key(old_ctx), None)   # (the header: fetch and destroy an engine's batches before reloading it)

    def error(self):
        return self._L.nsh_engine_error(self.h).decode()

    @property
    def ctx(self):
        return self._L.nsh_engine_ctx(self.h)

    @property
    def num_segments(self):
        return self._L.nsh_engine_num_segments(self.h)

    def segment_name(self, seg):
        return self._L.nsh_engine_segment_name(self.h, seg).decode()

    def segment_info(self, seg):
        n, a, p, t, b = C.c_uint32(), C.c_float(), C.c_uint64(), C.c_uint32(), C.c_int()
        rc = self._L.nsh_engine_segment_info(self.h, seg, C.byref(n), C.byref(a), C.byref(p), C.byref(t), C.byref(b))
        if rc != 0:
            raise IndexError(seg)
        return {"n_docs": n.value, "avgdl": a.value, "n_postings": p.value, "n_terms": t.value, "use_barrels": bool(b.value)}

    def segment_doc_len(self, seg):
        n = self.segment_info(seg)["n_docs"]
        p = self._L.nsh_engine_segment_doc_len(self.h, seg)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    def segment_postings(self, seg):
        nb = C.c_uint64()
        p = self._L.nsh_engine_segment_postings(self.h, seg, C.byref(nb))
        n = nb.value // 4
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n,)).copy().reshape(-1, 2) if n else np.zeros((0, 2), np.uint32)

    def lookup(self, seg, term):
        tid, df, cnt, off, idf = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_float()
        ok = self._L.nsh_engine_lookup(self.h, seg, term.encode(), C.byref(tid), C.byref(df), C.byref(cnt), C.byref(off), C.byref(idf))
        if not ok:
            return None
        return {"term_id": tid.value, "df": df.value, "count": cnt.value, "byte_off": off.value, "idf": idf.value}

    def build_refs(self, queries):
        Q = len(queries)
        qarr = _cstr_array(queries)
        qd = np.zeros(Q, dtype=QDESC_DTYPE)
        usable = np.zeros(Q, dtype=np.uint8)
        n = C.c_uint32(0)
        self._L.nsh_engine_build_refs(self.h, qarr, Q, qd.ctypes.data, None, 0, C.byref(n), usable.ctypes.data)
        refs = np.zeros(max(n.value, 1), dtype=TERM_DTYPE)
        rc = self._L.nsh_engine_build_refs(self.h, qarr, Q, qd.ctypes.data, refs.ctypes.data, n.value, C.byref(n), usable.ctypes.data)
        if rc != 0:
            raise RuntimeError("nsh_engine_build_refs failed")
        return qd, refs[: n.value], usable

    def doc_metadata(self, seg, doc):
        """The decorated fields of one document (None if it has no metadata.csv row)."""
        t, u, p, a = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_char_p()
        has = self._L.nsh_engine_doc_metadata(self.h, seg, doc, C.byref(t), C.byref(u), C.byref(p), C.byref(a))
        if not has:
            return None
        return {"title": t.value.decode(), "url": u.value.decode(), "publish_time": p.value.decode(), "author": a.value.decode()}

    def hits_to_json(self, query, k, has_found, found, hits):
        """Result assembly alone: JSON text for given hits (numpy array of HIT_DTYPE)."""
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        out = C.c_void_p()
        q = query.encode("utf-8") if isinstance(query, str) else query
        rc = self._L.nsh_engine_hits_to_json(self.h, q, k, 1 if has_found else 0, int(found), hits.ctypes.data, len(hits), C.byref(out))
        if rc != 0:
            raise RuntimeError("nsh_engine_hits_to_json failed")
        s = C.string_at(out).decode("utf-8")
        self._L.nsh_free(out)
        return s

    def search_batch_json(self, queries, k, decode=True):
        """Batch of searches straight to the /api/search JSON bodies."""
        Q = len(queries)
        offs = np.zeros(Q + 1, dtype=np.uint64)
        out = C.c_void_p()
        rc = self._L.nsh_engine_search_batch_json(self.h, _cstr_array(queries), Q, k, C.byref(out), offs.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"search_batch_json failed: {self.error()}")
        raw = C.string_at(out, int(offs[Q]))
        self._L.nsh_free(out)
        if not decode:
            return raw, offs
        return [raw[int(offs[q]):int(offs[q + 1])].decode("utf-8") for q in range(Q)]

    def search_json(self, query, k):
        out = C.c_void_p()
        rc = self._L.nsh_engine_search_json(self.h, query.encode(), k, C.byref(out))
        if rc != 0:
            raise RuntimeError(f"search failed: {self.error()}")
        s = C.string_at(out).decode()
        self._L.nsh_free(out)
        return s

    def search_batch(self, queries, k, flags=NS_FLAG_OR):
        Q, K = len(queries), clamp_k(k)
        hits = np.empty((Q, K), dtype=HIT_DTYPE)
        nhits = np.zeros(Q, dtype=np.uint32)
        found = np.zeros(Q, dtype=np.uint64)
        has_found = np.zeros(Q, dtype=np.uint8)
        rc = self._L.nsh_engine_search_batch(self.h, _cstr_array(queries), Q, k, flags, hits.ctypes.data,
                                             nhits.ctypes.data, found.ctypes.data, has_found.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"search_batch failed: {self.error()}")
        return hits, nhits, found, has_found

    def prepare(self, queries, k, flags=NS_FLAG_OR):
        b = C.c_void_p()
        rc = self._L.nsh_engine_prepare(self.h, _cstr_array(queries), len(queries), k, flags, C.byref(b))
        if rc != 0:
            raise RuntimeError(f"prepare failed: {self.error()}")
        import weakref
        bt = Batch(b, len(queries), clamp_k(k))
        _LIVE_BATCHES.setdefault(_ctx_key(self.ctx), weakref.WeakSet()).add(bt)
        return bt

    def set_cache(self, on):
        self._L.nsh_engine_set_cache(self.h, 1 if on else 0)

    def cache_size(self):
        return self._L.nsh_engine_cache_size(self.h)

    def semantic_info(self):
        rows, dim = C.c_uint32(), C.c_uint32()
        on = self._L.nsh_engine_semantic_info(self.h, C.byref(rows), C.byref(dim))
        return bool(on), rows.value, dim.value

    def semantic_row(self, row):
        """(term, fp32 vector) of one row of the loaded embedding table."""
        _, _, dim = self.semantic_info()
        t, v = C.c_char_p(), C.c_void_p()
        if self._L.nsh_engine_semantic_row(self.h, row, C.byref(t), C.byref(v)) != 0:
            raise IndexError(row)
        return t.value.decode(), np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_float)), shape=(dim,)).copy()

    def expand(self, query):
        """[(term, fp32 weight bits)] a search scores for this query, in scoring order."""
        out = C.c_void_p()
        if self._L.nsh_engine_expand(self.h, query.encode(), C.byref(out)) != 0:
            raise RuntimeError(f"expand failed: {self.error()}")
        text = C.string_at(out).decode()
        self._L.nsh_free(out)
        return [(ln.split("\t")[0], int(ln.split("\t")[1], 16)) for ln in text.splitlines()]

    def build_impacts(self):
        """Precompute every list's per-posting term scores on the device (optional second posting stream)."""
        if self._L.nsh_engine_build_impacts(self.h) != 0:
            raise RuntimeError(f"build_impacts failed: {self.error()}")

    def build_blockmax(self):
        """Block maxima for every list of >= 512 postings (ns_segment_build_blockmax); see use_pruning."""
        if self._L.nsh_engine_build_blockmax(self.h) != 0:
            raise RuntimeError(f"build_blockmax failed: {self.error()}")

    def use_pruning(self, on):
        """Single-term queries skip the blocks that cannot enter their top-K (found stays exact); off by default."""
        self._L.nsh_engine_use_pruning(self.h, 1 if on else 0)

    def use_merge(self, on):
        """Two-list groups: the merge body (default) or the driver-stream body."""
        self._L.nsh_engine_use_merge(self.h, 1 if on else 0)

    def share_scores(self, mode):
        """0 never, 1 (default) batches that name their lists often enough, 2 every batch that can (ns_ctx_share_scores)"""
        self._L.nsh_engine_share_scores(self.h, int(mode))

    def build_packed(self):
        """Build every segment's compressed, blocked posting stream on the device (optional)."""
        if self._L.nsh_engine_build_packed(self.h) != 0:
            raise RuntimeError(f"build_packed failed: {self.error()}")

    def use_packed(self, mode):
        """0 off; 1 packed docIds + tf, norms from the fp32 norm stream (default); 2 norms through the 16-bit norm index."""
        self._L.nsh_engine_use_packed(self.h, int(mode))

    def use_skips(self, on):
        """Searches walk the skip tables reload() built (default) or ignore them."""
        self._L.nsh_engine_use_skips(self.h, 1 if on else 0)

    def use_impacts(self, on):
        self._L.nsh_engine_use_impacts(self.h, 1 if on else 0)

    def set_tuning(self, variant=0, min_items=0, split_postings=0):
        rc = hip_lib().ns_set_tuning(self.ctx, variant, min_items, split_postings)
        if rc != NS_OK:
            raise RuntimeError(hip_lib().ns_last_error(self.ctx).decode())


def invert_segment(seg_dir, device=0):
    """The reference's `lexicon <SEGMENT_DIR>` step with the inversion on the device; returns a stats dict."""
    pairs, kept, ms, call_s, total_s = u64_(), u64_(), C.c_float(), C.c_double(), C.c_double()
    rc = host_lib().nsh_invert_segment(str(seg_dir).encode(), device, C.byref(pairs), C.byref(kept), C.byref(ms), C.byref(call_s), C.byref(total_s))
    if rc != 0:
        raise RuntimeError(f"invert_segment failed: {host_lib().nsh_invert_error().decode()}")
    return {"pairs": pairs.value, "kept": kept.value, "device_ms": ms.value, "call_s": call_s.value, "total_s": total_s.value}


def u64_():
    return C.c_uint64()


# Batches alive per device context (keyed by the ctx pointer): a batch must be destroyed BEFORE its ctx (it returns its
# blocks and pinned slots to it).  A test that fails between prepare() and close() keeps its Batch alive in the traceback
# while its `finally` closes the engine; the engine therefore closes what is left of its batches first (close_batches_of).
_LIVE_BATCHES = {}


def _ctx_key(ctx):
    return ctx.value if isinstance(ctx, C.c_void_p) else int(ctx) if ctx else 0


def close_batches_of(ctx):
    for b in list(_LIVE_BATCHES.pop(_ctx_key(ctx), ())):
        b.close()


def prepare_raw(ctx, qd, refs, k, flags=NS_FLAG_OR):
    """ns_batch_prepare on descriptor arrays that are already in the C-ABI's layout (numpy QDESC_DTYPE / TERM_DTYPE)."""
    import weakref
    b = C.c_void_p()
    rc = hip_lib().ns_batch_prepare(ctx, qd.ctypes.data, refs.ctypes.data if len(refs) else None, len(qd), int(k), flags, C.byref(b))
    if rc != NS_OK:
        raise RuntimeError("ns_batch_prepare: " + hip_lib().ns_last_error(ctx).decode())
    bt = Batch(b, len(qd), int(k))
    _LIVE_BATCHES.setdefault(_ctx_key(ctx), weakref.WeakSet()).add(bt)
    return bt


def pipelined_search(ctx, batches, k, flags=NS_FLAG_OR, out=None, timed=False, depth=2):
    """Host -> host search of a sequence of batches [(qd, refs), ...] with up to `depth` batches in flight on the one
    ctx: while the device scores batch i the host prepares and uploads batch i+1 (.. i+depth-1), and batch i's results
    are fetched as soon as they have landed (NS_RUN_FETCH).  Yields (hits, nhits, found, info) per batch, in order.
    `out`: optional list of preallocated (hits, nhits, found) triples, reused round-robin (any number >= 1: a triple
    is only written when its batch is fetched)."""
    from collections import deque

    def collect(pb, pi):
        o = out[pi % len(out)] if out is not None else (np.empty((pb.Q, k), dtype=HIT_DTYPE), np.empty(pb.Q, dtype=np.uint32), np.empty(pb.Q, dtype=np.uint64))
        pb.fetch_into(*o)
        inf = pb.info()
        pb.close()
        return o + (inf,)

    flight = deque()
    for i, (qd, refs) in enumerate(batches):
        b = prepare_raw(ctx, qd, refs, k, flags)
        b.run(timed=timed, fetch=True)
        flight.append((b, i))
        if len(flight) >= depth:
            yield collect(*flight.popleft())
    while flight:
        yield collect(*flight.popleft())


def search_batch_raw(ctx, qd, refs, k, flags=NS_FLAG_OR):
    """Direct call of the C-ABI's one-shot entry point with numpy descriptor arrays."""
    L = hip_lib()
    Q, K = len(qd), int(k)
    hits = np.empty((Q, max(K, 1)), dtype=HIT_DTYPE)
    nhits = np.zeros(Q, dtype=np.uint32)
    found = np.zeros(Q, dtype=np.uint64)
    qd = np.ascontiguousarray(qd, dtype=QDESC_DTYPE)
    refs = np.ascontiguousarray(refs, dtype=TERM_DTYPE)
    rc = L.ns_search_batch(ctx, qd.ctypes.data, refs.ctypes.data if len(refs) else None, Q, K, hits.ctypes.data,
                           nhits.ctypes.data, found.ctypes.data, flags)
    return rc, hits, nhits, found

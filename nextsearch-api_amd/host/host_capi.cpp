// C wrappers of include/nextsearch_host.h over nextsearch::Engine.
// No C++ exception crosses the boundary: every entry runs inside try/catch (std::filesystem errors, bad_alloc on a
// corrupt count, ...) and reports failure through its return value and nsh_engine_error().
#include "invert.hpp"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/nextsearch_host.h"
#include "engine.hpp"
#include "gen_index.hpp"
#include "textutil.hpp"

struct nsh_engine {
    nextsearch::Engine eng;
    std::string err;      // the wrapper's last message; written and read under err_mtx (callers may share an engine between threads)
    std::mutex err_mtx;
    explicit nsh_engine(int device) : eng(device) {}
    explicit nsh_engine(const std::vector<int>& devices) : eng(devices) {}
};

static void nsh_set_err(nsh_engine* e, const std::string& msg) {
    if (!e) return;
    std::lock_guard<std::mutex> l(e->err_mtx);
    e->err = msg;
}
static void nsh_note(nsh_engine* e, const char* where, const char* what) {
    nsh_set_err(e, std::string(where) + ": " + what);
}
#define NSH_CATCH(e, where, failval)                                                        \
    catch (const std::exception& ex_) { nsh_note((e), (where), ex_.what()); return failval; } \
    catch (...) { nsh_note((e), (where), "unknown exception"); return failval; }
#define NSH_CATCH_VOID(e, where)                                          \
    catch (const std::exception& ex_) { nsh_note((e), (where), ex_.what()); } \
    catch (...) { nsh_note((e), (where), "unknown exception"); }

static std::vector<std::string> to_vec(const char* const* qs, uint32_t n) {
    std::vector<std::string> v(n);
    for (uint32_t i = 0; i < n; i++) v[i] = qs[i] ? qs[i] : "";
    return v;
}

extern "C" int nsh_gen_index(const char* index_dir, uint32_t n_segments, uint32_t docs_per_segment, uint32_t vocab,
                             uint64_t seed, int legacy_layout, uint64_t* total_postings_out) {
    try {
        nsx::GenParams p;
        p.index_dir = index_dir;
        p.n_segments = n_segments;
        p.docs_per_segment = docs_per_segment;
        p.vocab = vocab;
        p.seed = seed;
        p.legacy_layout = legacy_layout != 0;
        nsx::GenStats st = nsx::generate_index(p);
        if (total_postings_out) *total_postings_out = st.total_postings;
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

extern "C" int nsh_engine_open(const char* index_dir, int device, nsh_engine** out) { try {
    if (!out) return -1;
    nsh_engine* e = new nsh_engine(device);
    e->eng.index_dir = index_dir ? index_dir : "";
    *out = e;
    if (!e->eng.reload()) { nsh_set_err(e, e->eng.last_error()); return -1; }
    return 0;
} NSH_CATCH(out ? *out : nullptr, "nsh_engine_open", -1)
}

extern "C" int nsh_engine_open_multi(const char* index_dir, const int* devices, uint32_t n_devices, nsh_engine** out) { try {
    if (!out || !devices || n_devices == 0) return -1;
    nsh_engine* e = new nsh_engine(std::vector<int>(devices, devices + n_devices));
    e->eng.index_dir = index_dir ? index_dir : "";
    *out = e;
    if (!e->eng.reload()) { nsh_set_err(e, e->eng.last_error()); return -1; }
    return 0;
} NSH_CATCH(out ? *out : nullptr, "nsh_engine_open_multi", -1)
}
extern "C" uint32_t nsh_engine_num_devices(nsh_engine* e) { try { return e ? (uint32_t)e->eng.num_devices() : 0; } NSH_CATCH(e, "nsh_engine_num_devices", 0)
}
extern "C" void nsh_shard_bounds(uint64_t n_queries, uint32_t r, uint32_t n, uint64_t* begin, uint64_t* end) {
    const auto b = nextsearch::Engine::shard_bounds((size_t)n_queries, r, n);
    if (begin) *begin = b.first;
    if (end) *end = b.second;
}

extern "C" void nsh_engine_close(nsh_engine* e) { delete e; }
// Engine::reload() (include/api_engine.hpp:65) on the directory given at open: 0 on success; on failure the engine
// keeps serving what it served before and nsh_engine_error() says why.
extern "C" int nsh_engine_reload(nsh_engine* e) { try {
    if (!e) return -1;
    if (!e->eng.reload()) { nsh_set_err(e, e->eng.last_error()); return -1; }
    return 0;
} NSH_CATCH(e, "nsh_engine_reload", -1)
}
extern "C" const char* nsh_engine_error(nsh_engine* e) { try {
    if (!e) return "null engine";
    // a copy per calling thread: the pointer stays valid until this thread asks again, whatever other threads do
    thread_local std::string mine;
    const std::string eng_err = e->eng.last_error();
    {
        std::lock_guard<std::mutex> l(e->err_mtx);
        if (!eng_err.empty()) e->err = eng_err;
        mine = e->err;
    }
    return mine.c_str();
} NSH_CATCH(e, "nsh_engine_error", "")
}
extern "C" ns_ctx* nsh_engine_ctx(nsh_engine* e) { try { return e ? e->eng.ctx() : nullptr;  } NSH_CATCH(e, "nsh_engine_ctx", nullptr)
}
extern "C" uint32_t nsh_engine_num_segments(nsh_engine* e) { try { return e ? (uint32_t)e->eng.segments.size() : 0;  } NSH_CATCH(e, "nsh_engine_num_segments", 0)
}
extern "C" const char* nsh_engine_segment_name(nsh_engine* e, uint32_t seg) { try {
    return (e && seg < e->eng.seg_names.size()) ? e->eng.seg_names[seg].c_str() : "";
} NSH_CATCH(e, "nsh_engine_segment_name", "")
}

// Result decoration (src/api_engine.cpp:516-531): the four fields of a document, valid until close/reload.
// Returns 1 if the document has a metadata row, else 0 (all four then point to "").
extern "C" int nsh_engine_doc_metadata(nsh_engine* e, uint32_t seg, uint32_t doc, const char** title, const char** url,
                                       const char** publish_time, const char** author) { try {
    static const char* kEmpty = "";
    const nsx::MetaFields* m = e ? e->eng.meta.get(seg, doc) : nullptr;
    if (title) *title = m ? m->title.c_str() : kEmpty;
    if (url) *url = m ? m->url.c_str() : kEmpty;
    if (publish_time) *publish_time = m ? m->publish_time.c_str() : kEmpty;
    if (author) *author = m ? m->author.c_str() : kEmpty;
    return m ? 1 : 0;
} NSH_CATCH(e, "nsh_engine_doc_metadata", -1)
}
// JSON text of one result assembled from given hits (the decoration + serialisation step alone; no device needed).
extern "C" int nsh_engine_hits_to_json(nsh_engine* e, const char* query, int k, int has_found, uint64_t found,
                                       const ns_hit* hits, uint32_t nhits, char** json_out) { try {
    if (!e || !json_out) return -1;
    nextsearch::SearchResult r;
    r.query = query ? query : "";
    r.k = std::max(1, std::min(k, 100));
    r.segments = (int)e->eng.segments.size();
    r.has_found = has_found != 0;
    r.found = found;
    if (nhits && !hits) return -1;
    for (uint32_t i = 0; i < nhits; i++) {
        if (hits[i].seg_id >= e->eng.segments.size()) { nsh_set_err(e, "nsh_engine_hits_to_json: hit names a segment that is not loaded"); return -1; }
        r.hits.push_back(nextsearch::SearchHit{hits[i].score, hits[i].seg_id, hits[i].doc_id});
    }
    const std::string js = e->eng.to_json(r);
    char* out = (char*)std::malloc(js.size() + 1);
    if (!out) return -1;
    std::memcpy(out, js.c_str(), js.size() + 1);
    *json_out = out;
    return 0;
} NSH_CATCH(e, "nsh_engine_hits_to_json", -1)
}

extern "C" int nsh_engine_segment_info(nsh_engine* e, uint32_t seg, uint32_t* n_docs, float* avgdl, uint64_t* n_postings,
                                       uint32_t* n_terms, int* use_barrels) { try {
    if (!e || seg >= e->eng.segments.size()) return -1;
    const auto& s = e->eng.segments[seg];
    if (n_docs) *n_docs = s.N;
    if (avgdl) *avgdl = s.avgdl;
    if (n_postings) *n_postings = s.postings_bytes / 8;
    if (n_terms) *n_terms = (uint32_t)s.lex.size();
    if (use_barrels) *use_barrels = s.use_barrels ? 1 : 0;
    return 0;
} NSH_CATCH(e, "nsh_engine_segment_info", -1)
}

extern "C" const uint32_t* nsh_engine_segment_doc_len(nsh_engine* e, uint32_t seg) { try {
    return (e && seg < e->eng.segments.size()) ? e->eng.segments[seg].doc_len.data() : nullptr;
} NSH_CATCH(e, "nsh_engine_segment_doc_len", nullptr)
}
extern "C" const void* nsh_engine_segment_postings(nsh_engine* e, uint32_t seg, uint64_t* nbytes) { try {
    if (!e || seg >= e->eng.segments.size()) return nullptr;
    const std::vector<uint8_t>* raw = e->eng.raw_postings(seg);   // read from the inverted files on first request
    if (!raw) return nullptr;
    if (nbytes) *nbytes = raw->size();
    return raw->data();
} NSH_CATCH(e, "nsh_engine_segment_postings", nullptr)
}

extern "C" int nsh_engine_lookup(nsh_engine* e, uint32_t seg, const char* term, uint32_t* term_id, uint32_t* df,
                                 uint32_t* count, uint64_t* byte_off, float* idf) { try {
    if (!e || seg >= e->eng.segments.size() || !term) return 0;
    const auto& s = e->eng.segments[seg];
    auto it = s.lex.find(term);
    if (it == s.lex.end()) return 0;
    const nsx::LexEntry& le = it->second;
    if (term_id) *term_id = le.termId;
    if (df) *df = le.df;
    if (count) *count = le.count;
    if (byte_off) *byte_off = s.list_byte_offset(le);
    if (idf) *idf = nextsearch::bm25_idf(s.N, le.df);
    return 1;
} NSH_CATCH(e, "nsh_engine_lookup", -1)
}

extern "C" float nsh_bm25_idf(uint32_t n_docs, uint32_t df) { return nextsearch::bm25_idf(n_docs, df); }   // arithmetic only

extern "C" uint32_t nsh_base_terms(const char* query, char* buf, uint32_t cap) { try {
    auto terms = nextsearch::base_terms(query ? query : "");
    std::string joined;
    for (size_t i = 0; i < terms.size(); i++) {
        if (i) joined.push_back(' ');
        joined += terms[i];
    }
    if (buf && cap) {
        size_t n = std::min<size_t>(joined.size(), cap - 1);
        std::memcpy(buf, joined.data(), n);
        buf[n] = 0;
    }
    return (uint32_t)terms.size();
} NSH_CATCH(nullptr, "nsh_base_terms", 0)
}

extern "C" int nsh_engine_build_refs(nsh_engine* e, const char* const* queries, uint32_t n_queries, ns_query_desc* qd,
                                     ns_term_ref* refs, uint32_t refs_cap, uint32_t* n_refs, uint8_t* usable) { try {
    if (!e) return -1;
    std::vector<ns_query_desc> q;
    std::vector<ns_term_ref> r;
    std::vector<uint8_t> u;
    e->eng.build_refs(to_vec(queries, n_queries), q, r, u);
    if (n_refs) *n_refs = (uint32_t)r.size();
    if (qd) std::memcpy(qd, q.data(), q.size() * sizeof(ns_query_desc));
    if (usable) std::memcpy(usable, u.data(), u.size());
    if (r.size() > refs_cap) return 1;
    if (refs && !r.empty()) std::memcpy(refs, r.data(), r.size() * sizeof(ns_term_ref));
    return 0;
} NSH_CATCH(e, "nsh_engine_build_refs", -1)
}

extern "C" int nsh_engine_search_json(nsh_engine* e, const char* query, int k, char** json_out) { try {
    if (!e || !json_out) return -1;
    std::string s;
    const bool ok = e->eng.search_text(query ? query : "", k, s);   // Engine::search: the result cache included
    if (!ok) { nsh_set_err(e, e->eng.last_error()); *json_out = nullptr; return -1; }
    *json_out = (char*)std::malloc(s.size() + 1);
    std::memcpy(*json_out, s.c_str(), s.size() + 1);
    return 0;
} NSH_CATCH(e, "nsh_engine_search_json", -1)
}

// Batch of searches to JSON bodies: *text_out receives all bodies back to back (free with nsh_free),
// offsets[q] .. offsets[q+1] delimit body q (offsets has n_queries + 1 entries).
extern "C" int nsh_engine_search_batch_json(nsh_engine* e, const char* const* queries, uint32_t n_queries, int k,
                                            char** text_out, uint64_t* offsets) { try {
    if (!e || !text_out || !offsets) return -1;
    std::vector<std::string> qs = to_vec(queries, n_queries), bodies;
    if (!e->eng.search_batch_json(qs, k, bodies)) { nsh_set_err(e, e->eng.last_error()); return -1; }
    size_t total = 0;
    for (auto& b : bodies) total += b.size();
    char* buf = (char*)std::malloc(total + 1);
    if (!buf) return -1;
    size_t pos = 0;
    for (uint32_t q = 0; q < n_queries; q++) {
        offsets[q] = pos;
        std::memcpy(buf + pos, bodies[q].data(), bodies[q].size());
        pos += bodies[q].size();
    }
    offsets[n_queries] = pos;
    buf[pos] = 0;
    *text_out = buf;
    return 0;
} NSH_CATCH(e, "nsh_engine_search_batch_json", -1)
}
extern "C" void nsh_free(void* p) { std::free(p); }

extern "C" int nsh_engine_search_batch(nsh_engine* e, const char* const* queries, uint32_t n_queries, int k, uint32_t flags,
                                       ns_hit* hits, uint32_t* nhits, uint64_t* found, uint8_t* has_found) { try {
    if (!e) return -1;
    // Engine::search_batch_flat: the caller's arrays are the outputs; whichever the caller left out is kept in scratch
    const uint32_t K = (uint32_t)std::max(1, std::min(k, 100));
    std::vector<nextsearch::Engine::QueryView> views(n_queries);
    for (uint32_t q = 0; q < n_queries; q++) views[q] = {queries[q] ? queries[q] : "", queries[q] ? std::strlen(queries[q]) : 0};
    std::vector<ns_hit> h_;
    std::vector<uint32_t> n_;
    std::vector<uint64_t> f_;
    std::vector<uint8_t> u_;
    if (!hits) { h_.resize((size_t)n_queries * K); hits = h_.data(); }
    if (!nhits) { n_.resize(n_queries); nhits = n_.data(); }
    if (!found) { f_.resize(n_queries); found = f_.data(); }
    if (!has_found) { u_.resize(n_queries); has_found = u_.data(); }
    if (!e->eng.search_batch_flat(views.data(), n_queries, k, flags, hits, nhits, found, has_found)) { nsh_set_err(e, e->eng.last_error()); return -1; }
    for (uint32_t q = 0; q < n_queries; q++) {
        if (has_found[q]) continue;   // the early return (src/api_engine.cpp:407): no hits, no found
        nhits[q] = 0; found[q] = 0;
        for (uint32_t i = 0; i < K; i++) hits[(size_t)q * K + i] = ns_hit{-__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu};
    }
    return 0;
} NSH_CATCH(e, "nsh_engine_search_batch", -1)
}

// The reference's `lexicon <SEGMENT_DIR>` tool with the inversion on the device (host/invert.hpp).
static thread_local std::string g_invert_err;
extern "C" const char* nsh_invert_error() { return g_invert_err.c_str(); }
extern "C" int nsh_invert_segment(const char* seg_dir, int device, uint64_t* pairs, uint64_t* kept, float* device_ms,
                                  double* call_s, double* total_s) { try {
    if (!seg_dir) return -1;
    ns_ctx* ctx = nullptr;
    if (ns_ctx_create(device, &ctx) != NS_OK) { g_invert_err = std::string("ns_ctx_create: ") + ns_last_error(nullptr); return -1; }
    nsx::InvertStats st;
    const bool ok = nsx::invert_segment(ctx, seg_dir, st, g_invert_err);
    ns_ctx_destroy(ctx);
    if (pairs) *pairs = st.pairs;
    if (kept) *kept = st.kept;
    if (device_ms) *device_ms = st.device_ms;
    if (call_s) *call_s = st.call_s;
    if (total_s) *total_s = st.total_s;
    return ok ? 0 : -1;
} NSH_CATCH(nullptr, "nsh_invert_segment", -1)
}

// Semantic expansion (src/api_engine.cpp:409-417): rows/dim of the loaded embedding table (0/0: none), and the
// weighted terms a query is scored with, one "term<TAB>fp32 weight bits in hex" line each, in scoring order.
extern "C" int nsh_engine_semantic_info(nsh_engine* e, uint32_t* rows, uint32_t* dim) { try {
    if (!e) return -1;
    if (rows) *rows = e->eng.sem.enabled ? (uint32_t)e->eng.sem.terms.size() : 0;
    if (dim) *dim = e->eng.sem.enabled ? (uint32_t)e->eng.sem.dim : 0;
    return e->eng.sem.enabled ? 1 : 0;
} NSH_CATCH(e, "nsh_engine_semantic_info", -1)
}
extern "C" int nsh_engine_semantic_row(nsh_engine* e, uint32_t row, const char** term, const float** vec) { try {
    if (!e || !e->eng.sem.enabled || row >= e->eng.sem.terms.size()) return -1;
    if (term) *term = e->eng.sem.terms[row].c_str();
    if (vec) *vec = e->eng.sem.vecs.data() + (size_t)row * (size_t)e->eng.sem.dim;
    return 0;
} NSH_CATCH(e, "nsh_engine_semantic_row", -1)
}
extern "C" int nsh_engine_expand(nsh_engine* e, const char* query, char** text_out) { try {
    if (!e || !query || !text_out) return -1;
    std::vector<nsx::WeightedTerms> w;
    if (!e->eng.expand_queries({std::string(query)}, w)) { nsh_set_err(e, e->eng.last_error()); return -1; }
    std::string o;
    char buf[16];
    for (const auto& tw : w[0]) {
        uint32_t bits;
        std::memcpy(&bits, &tw.second, 4);
        std::snprintf(buf, sizeof(buf), "%08x", bits);
        o += tw.first; o += '\t'; o += buf; o += '\n';
    }
    *text_out = (char*)std::malloc(o.size() + 1);
    if (!*text_out) return -1;
    std::memcpy(*text_out, o.c_str(), o.size() + 1);
    return 0;
} NSH_CATCH(e, "nsh_engine_expand", -1)
}

// Search-result cache of Engine::search (src/api_engine.cpp:190-250): on by default as in the reference.
extern "C" void nsh_engine_set_cache(nsh_engine* e, int on) { try { if (e) e->eng.set_cache(on != 0);  } NSH_CATCH_VOID(e, "nsh_engine_set_cache")
}
extern "C" uint32_t nsh_engine_cache_size(nsh_engine* e) { try { return e ? (uint32_t)e->eng.cache_size() : 0;  } NSH_CATCH(e, "nsh_engine_cache_size", 0)
}

extern "C" int nsh_engine_build_impacts(nsh_engine* e) { try {
    if (!e) return -1;
    if (!e->eng.build_impacts()) { nsh_set_err(e, e->eng.last_error()); return -1; }
    return 0;
} NSH_CATCH(e, "nsh_engine_build_impacts", -1)
}
extern "C" int nsh_engine_build_packed(nsh_engine* e) { try {
    if (!e) return -1;
    if (!e->eng.build_packed()) { nsh_set_err(e, e->eng.last_error()); return -1; }
    return 0;
} NSH_CATCH(e, "nsh_engine_build_packed", -1)
}
extern "C" void nsh_engine_use_packed(nsh_engine* e, int on) { try { if (e) e->eng.use_packed(on); } NSH_CATCH_VOID(e, "nsh_engine_use_packed")
}
extern "C" int nsh_engine_build_blockmax(nsh_engine* e) { try {
    if (!e) return -1;
    if (!e->eng.build_blockmax()) { nsh_set_err(e, e->eng.last_error()); return -1; }
    return 0;
} NSH_CATCH(e, "nsh_engine_build_blockmax", -1)
}
extern "C" void nsh_engine_use_pruning(nsh_engine* e, int on) { try { if (e) e->eng.use_pruning(on != 0); } NSH_CATCH_VOID(e, "nsh_engine_use_pruning")
}
extern "C" void nsh_engine_use_merge(nsh_engine* e, int on) { try { if (e) e->eng.use_merge(on != 0); } NSH_CATCH_VOID(e, "nsh_engine_use_merge")
}
extern "C" void nsh_engine_share_scores(nsh_engine* e, int mode) { try { if (e) e->eng.share_scores(mode); } NSH_CATCH_VOID(e, "nsh_engine_share_scores")
}
extern "C" void nsh_engine_use_skips(nsh_engine* e, int on) { try { if (e) e->eng.use_skips(on != 0);  } NSH_CATCH_VOID(e, "nsh_engine_use_skips")
}
extern "C" void nsh_engine_use_impacts(nsh_engine* e, int on) { try { if (e) e->eng.use_impacts(on != 0);  } NSH_CATCH_VOID(e, "nsh_engine_use_impacts")
}

extern "C" int nsh_engine_prepare(nsh_engine* e, const char* const* queries, uint32_t n_queries, int k, uint32_t flags,
                                  ns_batch** out) { try {
    if (!e || !out) return -1;
    if (!e->eng.prepare(to_vec(queries, n_queries), k, flags, out)) { nsh_set_err(e, e->eng.last_error()); return -1; }
    return 0;
} NSH_CATCH(e, "nsh_engine_prepare", -1)
}

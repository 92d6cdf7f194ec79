// Host facade mirroring the reference's cord19::Engine query interface
// (include/api_engine.hpp:23-91): reload() and search(query, k) -> JSON keep their names, argument
// meaning and error behaviour; search_batch() is additive (a batch == Q independent searches).
//
// What stays on the host (reference file:line):
//   tokenise / stop-word filter      include/textutil.hpp:13-37, src/api_engine.cpp:388-397
//   lexicon probe per segment        src/api_engine.cpp:454-458
//   bm25_idf via glibc logf          src/api_engine.cpp:45-47,461
//   JSON assembly                    src/api_engine.cpp:400-404,505-536
// What crosses the C-ABI (include/nextsearch_hip.h) to the MI355X kernels:
//   posting traversal, BM25 term scores, accumulation, top-k, found   src/api_engine.cpp:441-504
//
//   result decoration from metadata.csv (title, url, publish_time, author): src/api_engine.cpp:516-531
// Out of scope here: the three LRU caches, semantic expansion, autocomplete.  There is no CPU scoring path: without a device search*() fails.
#pragma once

#include <cstdint>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/nextsearch_hip.h"
#include "index_format.hpp"
#include "metadata.hpp"
#include "semantic.hpp"
#include "term_dict.hpp"
#include "../csrc/ns_forkjoin.hpp"

namespace nextsearch {

struct SearchHit {
    float score;
    uint32_t seg;
    uint32_t doc;
};

struct SearchResult {
    std::string query;
    int k = 0;               // clamped K
    int segments = 0;
    bool has_found = false;  // false on the early-return path (src/api_engine.cpp:407): no "found" key
    uint64_t found = 0;
    std::vector<SearchHit> hits;
};

// bm25_idf (src/api_engine.cpp:45-47): u32 subtraction first, then int->float, fp32 throughout.
float bm25_idf(uint32_t N, uint32_t df);

class Engine {
public:
    nsx::fs::path index_dir;
    std::vector<std::string> seg_names;
    std::vector<nsx::SegmentData> segments;
    nsx::MetadataTable meta;   // <index>/metadata.csv, parsed once at reload() (src/api_engine.cpp:110-113,:516-531)
    nsx::SemanticTable sem;    // optional embeddings (src/api_engine.cpp:115-153): when loaded, every search expands its terms (:409-417)
    // term -> per-segment {byte_off, count, idf}, built at reload() next to the lexicons (term_dict.hpp):
    // the one probe per query term that replaces the reference's per-(term, segment) seg.lex.find + bm25_idf (:454-461)
    nsx::TermDict dict;

    // device < 0: host-only (index + query preparation; every search call fails loudly)
    explicit Engine(int device = 0);
    // SURVEY.md 8(e), "one host thread + ns_ctx per GPU": the index is REPLICATED on every listed device (reload() uploads
    // it to each), and search_batch_flat / search_batch cut a batch into contiguous shards of ceil(Q / N) queries, one per
    // device, each driven by its own host thread through its own context; the shards' results land in the caller's one
    // set of host arrays (plain D2H per device: in one process no collective is needed — bench.py's multi-PROCESS form
    // keeps the RCCL all-gather).  Queries are independent (the reference serialises them behind one mutex,
    // src/api_engine.cpp:372), so the cut changes no result.  The same device may be listed twice (two contexts on one
    // GPU: what the one-GPU test box exercises).  devices[0] is the primary context: single searches, semantic
    // expansion's similarity search and ctx() use it.
    explicit Engine(const std::vector<int>& devices);
    size_t num_devices() const { return 1 + replicas_.size(); }
    // [begin, end) of shard r of n over Q queries: contiguous, ceil(Q / n) each, the last ones short or empty
    static std::pair<size_t, size_t> shard_bounds(size_t Q, size_t r, size_t n) {
        const size_t per = n ? (Q + n - 1) / n : Q;
        return {std::min(Q, r * per), std::min(Q, (r + 1) * per)};
    }
    ~Engine();
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    bool reload();                                              // include/api_engine.hpp:65
    // Optional: per-posting term scores for every list of every lexicon, built on the device
    // (ns_segment_build_impacts); searches then read {docId, score} instead of {docId, tf} + norm.  Same results.
    bool build_impacts();
    void use_impacts(bool on);
    // reload() builds skip tables for every segment's frequent lists (ns_segment_build_skips); off = searches ignore them
    void use_skips(bool on);
    // Optional: blocks of 256 postings with 8/16/32-bit docId offsets, 8-bit tf and a 16-bit norm index,
    // built on the device next to the raw stream (ns_segment_build_packed); the driver streams then read 4-7 B per posting
    // instead of 12.  Same results.
    bool build_packed();
    // Optional (block-max scores): per 256 postings of every list of >= 512 postings the largest term score,
    // built on the device (ns_segment_build_blockmax).  use_pruning(true): single-term queries then skip, unread, the blocks
    // that cannot enter their top-K; `found` stays exact (it is the list's posting count).  Same results; off by default.
    bool build_blockmax();
    void use_pruning(bool on);
    // groups of exactly two lists: the two-list merge body (default) or the driver-stream body like every other group
    void use_merge(bool on);
    // ns_ctx_share_scores on every device context: 0 never, 1 (default) batches that name their lists often enough, 2 always
    void share_scores(int mode);
    void use_packed(int mode);   // 0 off, 1 packed docIds + tf with the fp32 norm stream (default), 2 norms through the 16-bit index
    std::string search(const std::string& query, int k);        // include/api_engine.hpp:66 (JSON text, dump(2) layout)
    // Search-result cache around search() (src/api_engine.cpp:190-250,:380-385,:539): key "query|K", at most 2600
    // entries, least recently used evicted, a hit returns the stored body plus "from_cache": true.  In memory only:
    // the reference also rewrites search_cache.json in the CWD on every insert (:245-249), which is not reproduced.
    bool search_text(const std::string& query, int k, std::string& body);   // search() with the failure visible to the caller (body = the message then)
    static constexpr size_t kMaxCacheSize = 2600;               // include/api_engine.hpp:42
    bool search_hits(const std::string& query, int k, uint32_t flags, SearchResult& out);
    bool search_batch(const std::vector<std::string>& queries, int k, uint32_t flags, std::vector<SearchResult>& out);
    // The same batch with flat, caller-owned outputs in the C-ABI's layout — hits Q x K (unused tail entries {-inf, ~0, ~0}),
    // nhits[Q], found[Q], usable[Q] (0 = the early return of src/api_engine.cpp:407: no "found") — and no per-query
    // allocation.  A large batch is cut into sub-batches that are pipelined on the one device context: the host prepares
    // sub-batch i+1 (tokenise, dictionary probes) while the device scores sub-batch i (include/nextsearch_hip.h:
    // NS_RUN_FETCH).  A batch == Q independent searches, so the cut changes no result.
    struct QueryView { const char* p; size_t n; };
    bool search_batch_flat(const QueryView* queries, size_t Q, int k, uint32_t flags, ns_hit* hits, uint32_t* nhits,
                           uint64_t* found, uint8_t* usable);

    // Query preparation only: flattened term refs in the C-ABI's layout.
    // usable[q] == 0 marks the early-return case (no base terms, or no segments).
    // `expanded`: the queries' weighted terms from semantic expansion (nullptr: base terms, weight 1.0)
    void build_refs_range(const std::vector<std::string>& queries, size_t q0, size_t q1, std::vector<ns_query_desc>& qd,
                          std::vector<ns_term_ref>& refs, std::vector<uint8_t>& usable,
                          const std::vector<nsx::WeightedTerms>* expanded = nullptr) const;
    // The weighted query terms a search scores (base terms, or their semantic expansion when embeddings are loaded)
    bool expand_queries(const std::vector<std::string>& queries, std::vector<nsx::WeightedTerms>& out) const;
    void build_refs(const std::vector<std::string>& queries, std::vector<ns_query_desc>& qd,
                    std::vector<ns_term_ref>& refs, std::vector<uint8_t>& usable) const;
    // Staged form used by bench.py: descriptors resident on the device, caller drives ns_batch_*.
    bool prepare(const std::vector<std::string>& queries, int k, uint32_t flags, ns_batch** out);

    std::string to_json(const SearchResult& r) const;
    std::string to_json_impl(const SearchResult& r) const;
    // A batch of searches straight to the /api/search JSON bodies (result assembly on several host threads).
    bool search_batch_json(const std::vector<std::string>& queries, int k, std::vector<std::string>& out);
    ns_ctx* ctx() const { return ctx_; }
    std::string last_error() const { std::lock_guard<std::recursive_mutex> lock(mtx_); return err_; }
    void set_cache(bool on) { std::lock_guard<std::recursive_mutex> lock(mtx_); cache_on_ = on; if (!on) { cache_.clear(); lru_.clear(); } }
    size_t cache_size() const { std::lock_guard<std::recursive_mutex> lock(mtx_); return cache_.size(); }
    // The raw posting payload of a segment in host memory, read from the inverted files on first request (tests and
    // tools; the engine itself never holds it: reload() streams the files to the device from their mappings).
    const std::vector<uint8_t>* raw_postings(uint32_t seg);

private:
    // The reference's Engine::mtx (include/api_engine.hpp:33, taken at src/api_engine.cpp:54,:168,:372): one lock around
    // every entry that touches the cache, the error string, the device context or the loaded index.
    mutable std::recursive_mutex mtx_;   // recursive: public entries call each other (search -> build_refs -> expand_queries)
    bool search_batch_locked(const std::vector<std::string>& queries, int k, uint32_t flags, std::vector<SearchResult>& out);
    bool search_hits_locked(const std::string& query, int k, uint32_t flags, SearchResult& out);
    std::vector<std::unique_ptr<std::vector<uint8_t>>> raw_postings_;
    void release_device_segments();
    struct CacheEntry { std::string body; std::list<std::string>::iterator lru; };
    std::unordered_map<std::string, CacheEntry> cache_;
    std::list<std::string> lru_;   // most recently used at the front
    bool cache_on_ = true;
    // query preparation of queries [q0, q1) through the term dictionary: refs appended to `refs`, qd[q - q0] filled with
    // term_begin relative to `refs`' start, usable[q - q0] set
    void build_refs_views(const QueryView* queries, size_t q0, size_t q1, ns_query_desc* qd, std::vector<ns_term_ref>& refs,
                          uint8_t* usable, std::vector<char>& scratch, std::vector<uint32_t>& gids) const;
    // host threads of query preparation (kept from batch to batch) and their scratch
    struct PrepScratch { std::vector<ns_term_ref> refs; std::vector<char> text; std::vector<uint32_t> gids; };
    mutable std::unique_ptr<ForkJoin> pool_;
    mutable std::vector<PrepScratch> scratch_;
    std::vector<ns_query_desc> flat_qd_;     // search_batch_flat's descriptor buffers, kept from call to call
    std::vector<ns_term_ref> flat_refs_;
    unsigned prep_width(size_t Q) const;
    void build_refs_parallel(const QueryView* queries, size_t q0, size_t q1, std::vector<ns_query_desc>& qd,
                             std::vector<ns_term_ref>& refs, uint8_t* usable) const;
    mutable bool refs_failed_ = false;   // build_refs could not run the device part of the expansion (err_ says why)
    int device_;
    ns_ctx* ctx_ = nullptr;
    std::vector<ns_seg*> dev_segs_;
    // further devices holding a replica of the index (multi-device engine): context + segments each
    struct Replica { int device = 0; ns_ctx* ctx = nullptr; std::vector<ns_seg*> segs; };
    std::vector<Replica> replicas_;
    // one contiguous range of a batch on one context: sub-batches pipelined (prepare(i+1) || kernels(i) || results(i-1))
    bool run_range(ns_ctx* ctx, const QueryView* queries, size_t q0, size_t q1, int K, uint32_t flags, ns_hit* hits, uint32_t* nhits,
                   uint64_t* found, uint8_t* usable, bool pooled_prep, std::string& err);
    mutable std::string err_;
};

}  // namespace nextsearch

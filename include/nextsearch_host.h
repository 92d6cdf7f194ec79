/*
 * nextsearch_host.h — C wrappers over the C++ host facade (nextsearch::Engine, the mirror of the
 * reference's cord19::Engine, include/api_engine.hpp:23-91) so that Python tests and bench.py can
 * drive it through ctypes.  The compute entry points all go through include/nextsearch_hip.h; this
 * header adds no scoring code and no CPU fallback.
 */
#ifndef NEXTSEARCH_HOST_H
#define NEXTSEARCH_HOST_H

#include <stdint.h>

#include "nextsearch_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nsh_engine nsh_engine;

/* Deterministic synthetic index in the reference's on-disk format. */
int nsh_gen_index(const char* index_dir, uint32_t n_segments, uint32_t docs_per_segment, uint32_t vocab,
                  uint64_t seed, int legacy_layout, uint64_t* total_postings_out);

/* Engine::reload() on index_dir.  device >= 0: create an ns_ctx on that GPU and upload every
 * segment; device < 0: host-only (index + query preparation; search calls fail).  On failure
 * returns non-zero and *out still receives an engine whose nsh_engine_error() explains why
 * (free it with nsh_engine_close). */
int  nsh_engine_open(const char* index_dir, int device, nsh_engine** out);
/* The multi-device engine ("one host thread + ns_ctx per GPU"): the index is replicated on every device
 * of devices[0 .. n_devices) (the same device may be listed twice: two contexts on one GPU), and nsh_engine_search_batch
 * cuts a batch into contiguous shards of ceil(Q / n_devices) queries, one per device, each driven by its own host thread;
 * the results land in the caller's one set of arrays.  devices[0] is the primary context (nsh_engine_ctx, single searches).
 * Same failure convention as nsh_engine_open. */
int  nsh_engine_open_multi(const char* index_dir, const int* devices, uint32_t n_devices, nsh_engine** out);
uint32_t nsh_engine_num_devices(nsh_engine* e);
/* [begin, end) of shard r of n over n_queries queries, as nsh_engine_search_batch cuts them (arithmetic only) */
void nsh_shard_bounds(uint64_t n_queries, uint32_t r, uint32_t n, uint64_t* begin, uint64_t* end);
void nsh_engine_close(nsh_engine* e);
/* Engine::reload() again on the same directory.  0 on success.  On failure (non-zero) the engine keeps the index,
 * the device copy and the caches it had (the reference swaps its segments in only after every one loaded,
 * src/api_engine.cpp:76-90).  On success the device context is a NEW one: re-read nsh_engine_ctx(). */
int  nsh_engine_reload(nsh_engine* e);
/* The engine's last failure message.  The pointer stays valid for the CALLING thread until it asks again (a copy per thread). */
const char* nsh_engine_error(nsh_engine* e);
ns_ctx* nsh_engine_ctx(nsh_engine* e);

/* The reference's `lexicon <SEGMENT_DIR>` tool (src/lexicon.cpp) with the inversion on the device
 * (ns_invert_forward): reads <seg>/terms.bin + forward.bin, writes barrels.bin, lexicon_bNNN.bin and
 * inverted_bNNN.bin byte for byte as the reference does.  0 on success; nsh_invert_error() otherwise. */
int nsh_invert_segment(const char* seg_dir, int device, uint64_t* pairs, uint64_t* kept, float* device_ms,
                       double* call_s, double* total_s);
const char* nsh_invert_error(void);

/* Search-result cache around nsh_engine_search_json (src/api_engine.cpp:190-250,:380-385,:539): "query|K" keys,
 * 2600 entries, LRU eviction, hits carry "from_cache": true.  On by default, in memory only; the batch entry
 * points never use it. */
void nsh_engine_set_cache(nsh_engine* e, int on);
uint32_t nsh_engine_cache_size(nsh_engine* e);

/* Semantic query expansion (src/api_engine.cpp:115-153,:409-417; src/semantic_embedding.cpp): reload() loads
 * <index>/embeddings.vec|embeddings.txt|glove.txt|vectors.txt (or $EMBEDDINGS_PATH) for the lexicons' terms and
 * uploads the table; every search then scores the expanded, weighted terms (similarity search on the device,
 * ns_sem_topk).  nsh_engine_semantic_info: 1 if a table is loaded (+ rows, dim).  nsh_engine_expand: the weighted
 * terms of one query, "term<TAB>%08x weight bits" per line in scoring order; free with nsh_free. */
int nsh_engine_semantic_info(nsh_engine* e, uint32_t* rows, uint32_t* dim);
int nsh_engine_expand(nsh_engine* e, const char* query, char** text_out);
/* Row `row` of the loaded table: its term and its `dim` L2-normalised values (valid until close/reload); -1 if absent. */
int nsh_engine_semantic_row(nsh_engine* e, uint32_t row, const char** term, const float** vec);

/* Optional impact streams for every list of every loaded lexicon (include/nextsearch_hip.h:
 * ns_segment_build_impacts / ns_ctx_use_impacts).  Not part of reload(): 8 B of HBM per posting. */
int  nsh_engine_build_impacts(nsh_engine* e);
void nsh_engine_use_impacts(nsh_engine* e, int on);
/* reload() builds skip tables for the frequent lists of every segment (ns_segment_build_skips); on = 0: searches ignore them. */
void nsh_engine_use_skips(nsh_engine* e, int on);
/* Optional packed posting streams for every loaded segment (include/nextsearch_hip.h: ns_segment_build_packed /
 * ns_ctx_use_packed): 4-7 B read per posting instead of 12, same results.  Not part of reload(): 8 B of HBM per posting. */
int  nsh_engine_build_packed(nsh_engine* e);
void nsh_engine_use_packed(nsh_engine* e, int on);
/* Optional block maxima for every list of >= 512 postings of every loaded segment (include/nextsearch_hip.h:
 * ns_segment_build_blockmax) and the switch for `found`-exact pruning of single-term queries (ns_ctx_use_pruning; off by
 * default).  Same results either way. */
int  nsh_engine_build_blockmax(nsh_engine* e);
void nsh_engine_use_pruning(nsh_engine* e, int on);
/* on = 0: two-list groups take the driver-stream body instead of the merge body (ns_ctx_use_merge; default 1).  Same results. */
void nsh_engine_use_merge(nsh_engine* e, int on);
/* Shared term scores (include/nextsearch_hip.h: ns_ctx_share_scores) on every device context of the engine: 0 never, 1 (default)
 * batches that name each distinct list often enough compute its BM25 term scores once per run, 2 every batch that can.  Same results. */
void nsh_engine_share_scores(nsh_engine* e, int mode);

/* ---- inspection (tests and tools).  NOT reload-safe: these accessors read the loaded index without taking the engine
 * lock and hand out pointers into it, and nsh_engine_reload() replaces that index — do not call them, or use what they
 * returned, while another thread reloads.  (The search and query-preparation entries above do take the lock, as every
 * entry of the reference's engine takes Engine::mtx.)  Likewise an ns_batch obtained through nsh_engine_prepare belongs
 * to the device context it was prepared on: fetch and destroy it before reloading. */
uint32_t nsh_engine_num_segments(nsh_engine* e);
const char* nsh_engine_segment_name(nsh_engine* e, uint32_t seg);
int nsh_engine_segment_info(nsh_engine* e, uint32_t seg, uint32_t* n_docs, float* avgdl, uint64_t* n_postings,
                            uint32_t* n_terms, int* use_barrels);
/* Result decoration from <index>/metadata.csv (src/api_engine.cpp:516-531, src/api_metadata.cpp): the
 * decorated fields of one document, valid until close/reload; 1 if the document has a metadata row. */
int nsh_engine_doc_metadata(nsh_engine* e, uint32_t seg, uint32_t doc, const char** title, const char** url,
                            const char** publish_time, const char** author);
/* Result assembly alone (src/api_engine.cpp:400-404,:505-536): JSON text for given hits; free with nsh_free. */
int nsh_engine_hits_to_json(nsh_engine* e, const char* query, int k, int has_found, uint64_t found,
                            const ns_hit* hits, uint32_t nhits, char** json_out);
/* Host copies of what gets uploaded (valid until close/reload). */
const uint32_t* nsh_engine_segment_doc_len(nsh_engine* e, uint32_t seg);
const void* nsh_engine_segment_postings(nsh_engine* e, uint32_t seg, uint64_t* nbytes);
/* Lexicon probe (src/api_engine.cpp:454-461).  Returns 1 if found, 0 if absent. */
int nsh_engine_lookup(nsh_engine* e, uint32_t seg, const char* term, uint32_t* term_id, uint32_t* df, uint32_t* count,
                      uint64_t* byte_off, float* idf);

float nsh_bm25_idf(uint32_t n_docs, uint32_t df);
/* Tokenise + filter (include/textutil.hpp:13-37, src/api_engine.cpp:391-397): writes the kept terms
 * separated by single spaces into buf (NUL-terminated, truncated to cap); returns the term count. */
uint32_t nsh_base_terms(const char* query, char* buf, uint32_t cap);

/* Query preparation only: fills qd[n_queries], usable[n_queries] and up to refs_cap refs;
 * *n_refs receives the number needed.  Returns 0, or 1 if refs_cap was too small. */
int nsh_engine_build_refs(nsh_engine* e, const char* const* queries, uint32_t n_queries, ns_query_desc* qd,
                          ns_term_ref* refs, uint32_t refs_cap, uint32_t* n_refs, uint8_t* usable);

/* Engine::search(query, k) -> JSON text with the reference's keys; caller frees with nsh_free. */
int  nsh_engine_search_json(nsh_engine* e, const char* query, int k, char** json_out);
void nsh_free(void* p);
/* A batch of searches straight to the /api/search JSON bodies (result assembly on several host threads):
 * *text_out receives all bodies back to back (free with nsh_free); offsets[q] .. offsets[q+1] delimit body q
 * (offsets has n_queries + 1 entries). */
int  nsh_engine_search_batch_json(nsh_engine* e, const char* const* queries, uint32_t n_queries, int k,
                                  char** text_out, uint64_t* offsets);
/* Batch search through ns_search_batch.  hits: n_queries*K (K = clamp(k,1,100)). */
int nsh_engine_search_batch(nsh_engine* e, const char* const* queries, uint32_t n_queries, int k, uint32_t flags,
                            ns_hit* hits, uint32_t* nhits, uint64_t* found, uint8_t* has_found);
/* Staged: query prep on the host, descriptors to the device; drive the result with ns_batch_*. */
int nsh_engine_prepare(nsh_engine* e, const char* const* queries, uint32_t n_queries, int k, uint32_t flags,
                       ns_batch** out);

#ifdef __cplusplus
}
#endif
#endif /* NEXTSEARCH_HOST_H */

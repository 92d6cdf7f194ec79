/*
 * nextsearch_hip.h — C-ABI of libnextsearch_hip.so: the MI355X (gfx950) posting-traversal /
 * BM25-scoring / top-k hot path of NextSearch's cord19::Engine::search.
 *
 * The reference has no FFI seam: the path is an inline loop inside Engine::search.  The boundary is
 * therefore cut where that loop begins and ends (reference file:line, /root/reference/...):
 *
 *   enter : src/api_engine.cpp:441   (qterms_w final; per segment: lexicon probe :454-458 and
 *                                     bm25_idf :45-47,:461 stay on the HOST and arrive here as
 *                                     ns_term_ref{byte_off,count,idf,qweight})
 *   leave : src/api_engine.cpp:504-505 (hits sorted by score + total_found)
 *
 * Everything is plain-old-data; no exceptions cross the boundary; every entry point returns
 * NS_OK (0) or a negative NS_E* code and leaves a message retrievable with ns_last_error().
 * There is NO CPU fallback: without a usable HIP device ns_ctx_create fails with NS_E_NODEVICE.
 *
 * Threading: an ns_ctx is bound to one device and one stream and is not re-entrant (the reference
 * serialises every engine entry behind Engine::mtx, src/api_engine.cpp:372).  Segments are
 * immutable after upload.  Use one ctx per host thread / per GPU.
 */
#ifndef NEXTSEARCH_HIP_H
#define NEXTSEARCH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NS_OK            0
#define NS_E_INVAL      -1   /* bad argument (null pointer, K out of range, offset outside segment ...) */
#define NS_E_NODEVICE   -2   /* no HIP device / HIP runtime failure at init */
#define NS_E_HIP        -3   /* a HIP call failed; see ns_last_error */
#define NS_E_NOMEM      -4
#define NS_E_STATE      -5   /* call sequence error (e.g. fetch before run) */

#define NS_MAX_K        100u /* src/api_engine.cpp:377: K = clamp(k,1,100) */

/* flags for ns_search_batch / ns_batch_prepare */
#define NS_FLAG_OR      0u   /* reference semantics: every touched doc is a candidate (src/api_engine.cpp:449-492) */
#define NS_FLAG_AND     1u   /* extension (BASELINE config 2): keep docs matched by every term ref of their segment */
#define NS_INFO_IMPACTS 0x100u /* ns_batch_info.flags only (output): the batch reads impact streams (ns_segment_build_impacts) */
#define NS_INFO_PACKED  0x200u /* ns_batch_info.flags only (output): the batch's driver streams read the packed posting blocks (ns_segment_build_packed) */
#define NS_INFO_SHARED  0x800u /* ns_batch_info.flags only (output): the batch computes the BM25 term scores of its distinct lists once per run (ns_ctx_share_scores) */
#define NS_INFO_PRUNED  0x400u /* ns_batch_info.flags only (output): some single-term queries of the batch skip posting blocks by their block maxima (ns_ctx_use_pruning) */

typedef struct ns_ctx   ns_ctx;
typedef struct ns_seg   ns_seg;
typedef struct ns_batch ns_batch;

/* One scored (query term, segment) pair — replaces the reference's
 * {seg.lex.find(term) -> LexEntry; bm25_idf(seg.N,e.df); seekg(e.offset)} triple
 * (src/api_engine.cpp:454-470).  Refs of one query are contiguous and in QUERY-TERM ORDER; refs of
 * different segments may interleave (their relative order per segment is what matters: it is the
 * fp32 accumulation order of src/api_engine.cpp:480). */
typedef struct ns_term_ref {
    uint32_t seg_id;     /* id given to ns_segment_upload */
    uint32_t count;      /* LexEntry.count: postings in the list (include/api_types.hpp:27) */
    uint64_t byte_off;   /* byte offset of the list inside the uploaded posting buffer (multiple of 8) */
    float    idf;        /* host-computed bm25_idf(N, df) (src/api_engine.cpp:45-47) */
    float    qweight;    /* 1.0f, or the semantic-expansion weight (src/api_engine.cpp:410-421) */
} ns_term_ref;

typedef struct ns_query_desc {
    uint32_t term_begin; /* first ns_term_ref of this query */
    uint32_t term_count; /* may be 0: such a query returns nhits = 0, found = 0 */
} ns_query_desc;

/* struct Hit {float s; uint32_t segId; uint32_t docId;} (src/api_engine.cpp:427-431) */
typedef struct ns_hit {
    float    score;
    uint32_t seg_id;
    uint32_t doc_id;
} ns_hit;

typedef struct ns_batch_info {
    uint64_t postings;        /* P = sum over term refs of count */
    uint64_t algo_bytes;      /* 8 * P: algorithmic bytes of one run */
    uint32_t n_queries;
    uint32_t n_items;         /* (query, segment, doc-range) work items == workgroups of the scoring kernel */
    uint32_t n_term_refs;
    uint32_t tile_docs;       /* docs per LDS accumulator tile */
    uint32_t k;
    uint32_t flags;
    float    last_score_kernel_ms; /* HIP-event time of the scoring kernel in the last timed run, <0 if none */
    float    last_total_ms;        /* HIP-event time of all kernels of the last timed run, <0 if none */
    uint32_t timed_runs;           /* timed runs accumulated since prepare (read back at every sync) */
    uint32_t shared_lists;         /* distinct posting lists whose term scores the batch computes once per run (0: it does not share) */
    double   sum_score_kernel_ms;  /* sum over timed runs of the scoring kernel's HIP-event time (with the shared-score kernel in front of it, if any) */
    double   sum_total_ms;         /* sum over timed runs of first-kernel-start .. last-kernel-end */
    uint64_t shared_postings;      /* postings of those lists */
} ns_batch_info;

/* ---- context ------------------------------------------------------------------------------ */
int  ns_ctx_create(int device, ns_ctx** out);
void ns_ctx_destroy(ns_ctx* ctx);
/* Use an externally owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) for all work
 * of this ctx; NULL restores the ctx's own stream. */
int  ns_ctx_set_stream(ns_ctx* ctx, void* hip_stream);
/* Message of the last failing call on this ctx (ctx == NULL: last failing ns_ctx_create of the
 * calling thread).  Never NULL. */
const char* ns_last_error(ns_ctx* ctx);
/* "gfx950 ..." device string of the ctx's device */
const char* ns_device_name(ns_ctx* ctx);

/* ---- segments (replaces the reference's open ifstreams: include/api_types.hpp:54-59) ------- */
/* Copies doc_len[N] and the flattened posting payload (all inverted_bNNN.bin back to back, or
 * inverted.bin) to HBM through a pinned staging buffer, once; precomputes the per-doc BM25 norm
 * k1*((1-b) + b*(dl/avgdl)) in fp32 with the reference's operation order (src/api_engine.cpp:478).
 * Host buffers stay owned by the caller and may be freed on return. */
int ns_segment_upload(ns_ctx* ctx, uint32_t seg_id, uint32_t n_docs, float avgdl,
                      const uint32_t* doc_len, const void* postings, uint64_t nbytes, ns_seg** out);
int ns_segment_release(ns_ctx* ctx, ns_seg* seg);
/* The same upload for a posting payload that is NOT one host buffer.  The reference keeps a segment's postings in
 * up to 64 inverted_bNNN.bin files and reads them through open streams (include/api_types.hpp:54-59,
 * src/api_segment.cpp:70-102); the host maps those files one after the other and appends each mapping, so the
 * payload never exists in host memory as a whole.  begin: reserves HBM for total_nbytes of postings and takes doc_len;
 * append: the next nbytes of the payload, in payload order (any multiple of 8; copied through the pinned staging
 * buffers before the call returns); end: all bytes must have arrived; publishes the segment under seg_id.
 * ns_segment_release abandons an upload that was begun but not ended. */
int ns_segment_upload_begin(ns_ctx* ctx, uint32_t seg_id, uint32_t n_docs, float avgdl, const uint32_t* doc_len,
                            uint64_t total_nbytes, ns_seg** out);
int ns_segment_upload_append(ns_ctx* ctx, ns_seg* seg, const void* bytes, uint64_t nbytes);
int ns_segment_upload_end(ns_ctx* ctx, ns_seg* seg);

/* Optional second posting stream of a segment (a format loaded NEXT TO the reference's).
 * For every list given here the device stores {u32 docId, f32 term score} per posting, index-aligned with the
 * uploaded {docId, tf} stream, where term score = (idf * (tf * (k1 + 1))) / (tf + k1*((1-b) + b*dl/avgdl)) —
 * src/api_engine.cpp:477-479 evaluated ONCE per posting with the same fp32 operations, instead of once per
 * posting per query.  A batch reads the impact stream when EVERY term ref in it names a registered list with
 * the bit-identical idf (otherwise the whole batch takes the {docId, tf} path); results are bit-identical
 * either way.  Costs 8 B of HBM per posting of the segment.  byte_off/counts as in ns_term_ref; lists must
 * not overlap.  May be called again to add lists or to replace a list's idf. */
int ns_segment_build_impacts(ns_ctx* ctx, ns_seg* seg, const uint64_t* byte_off, const uint32_t* counts,
                             const float* idfs, uint32_t n_lists);
/* on != 0: batches prepared from now on alternate between the ctx's stream and a second one, so that the first work
 * items of batch i+1 fill the wave slots the draining tail of batch i leaves idle (matters most for small batches:
 * a 2048-query shard of a strong-scaled batch).  The alternation applies to the ctx's own stream (a caller who passes
 * streams with ns_ctx_set_stream alternates them itself).  A batch's upload, kernels and result copy all stay on the one
 * stream it was prepared on; with overlap on, the descriptor upload is pulled by a kernel instead of the DMA engine (copies
 * of all streams share one in-order DMA queue, which would chain batch i+1's upload to batch i's result copy). */
int ns_ctx_set_overlap(ns_ctx* ctx, int on);
/* Host threads ns_batch_prepare may use for a large batch (regrouping the term refs, cutting work items, writing the
 * descriptors): 0 = automatic (up to 8, one per ~1500 queries), 1 = the calling thread only.  The prepared batch — every
 * descriptor byte and the launch order — does not depend on this number. */
int ns_ctx_set_host_threads(ns_ctx* ctx, uint32_t n);
/* Compressed, blocked posting stream of a segment, loaded NEXT TO the reference's raw format
 * (src/lexicon.cpp:104-128: {u32 docId, u32 tf} per posting) and built from it on the device, once:
 *   blocks of 256 postings; per block a base docId and a width code; per posting a docId offset of 8, 16 or 32 bits
 *   (frame of reference: whatever the block's doc span needs), tf in 8 bits (255 = escape to the raw stream) and a
 *   16-bit index into the segment's table of DISTINCT BM25 norms — the norm is a function of doc_len alone, so the
 *   index loses nothing; the per-posting fp32 norm stream (4 B) is not read at all.
 * 4 B (dense lists), 5 B or 7 B per posting are read instead of 12; results are bit-identical (same operations on the
 * same values; every test runs both ways).  `found` (src/api_engine.cpp:495) needs every posting visited, so this is
 * compression, not skipping.  The scoring kernel's DRIVER streams read the packed blocks (the bulk of the bytes);
 * foreign windows and doc tiles keep reading the raw stream.  Fails with NS_E_INVAL for a segment with more than
 * 65536 distinct document lengths.  Costs 8 B of HBM per posting (fixed 2 KB stride per block). */
int ns_segment_build_packed(ns_ctx* ctx, ns_seg* seg);
/* mode 0: batches prepared from now on ignore packed streams.  mode 1 (default): a batch whose segments all have one
 * reads docIds and tf from the packed blocks and the norms from the per-posting fp32 norm stream (6-7 B per posting, one
 * memory latency per round).  mode 2: the norms come through the blocks' 16-bit norm index instead (4-5 B per posting, but
 * the table look-up is a second, dependent access per round). */
int ns_ctx_use_packed(ns_ctx* ctx, int mode);
/* on = 0: batches prepared from now on ignore impact streams, and do not share term scores either (ns_ctx_share_scores builds
 * into the same per-segment buffer) (default: on = 1). */
int ns_ctx_use_impacts(ns_ctx* ctx, int on);
/* Skip tables (block metadata next to the reference's raw posting format, src/lexicon.cpp:104-128,
 * which has none: the reference walks every list from its first posting, src/api_engine.cpp:470-481).  For every list
 * given here the device stores, per 1024-doc cell of the segment's doc space, the index of the list's first posting in
 * that cell (4 B per cell and list, built from the uploaded postings on the device).  Term groups that are scored in
 * doc tiles (several frequent lists) then walk those cells and take exactly a list's postings of the cell instead of
 * estimating how many to load and searching the cell's end by docId; every posting is still visited (`found`,
 * src/api_engine.cpp:495) and results are bit-identical either way.  byte_off/counts as in ns_term_ref; a list that
 * is not docId-ascending keeps no table (its groups take the cursor path); lists given again are left as they are.
 * Meant for the frequent lists of a segment (the facade registers lists of >= n_docs / 512 postings at reload). */
int ns_segment_build_skips(ns_ctx* ctx, ns_seg* seg, const uint64_t* byte_off, const uint32_t* counts, uint32_t n_lists);
/* on = 0: batches prepared from now on ignore skip tables (default: on = 1). */
int ns_ctx_use_skips(ns_ctx* ctx, int on);
/* Block-max scores with `found`-exact pruning.  The reference reads every posting of every scored list
 * (src/api_engine.cpp:470-481) because `found` (:495) is the size of the union of the lists' docs.  For a query whose term
 * group in a segment is ONE list that size needs no reading: it is the number of the list's postings (every posting is a
 * doc of its own), and the group's top-K (:485-492) only needs the blocks whose best score can still enter it.
 * ns_segment_build_blockmax stores, for every list given, per 256 postings of the LIST the largest term score
 * (idf * (tf * (k1 + 1))) / (tf + k1*((1-b) + b*dl/avgdl)) — src/api_engine.cpp:477-479 with the given idf, the same fp32
 * operations as the scoring kernels (4 B per 256 postings, built on the device from the uploaded postings).  With
 * ns_ctx_use_pruning(ctx, 1) (default 0: every posting is read, as the reference does, and that is what bench.py's `value`
 * and `roofline` measure) a single-term group whose list is registered with the bit-identical idf and whose weight is
 * positive visits its blocks in docId order and skips, unread, every block whose maximum times the weight is <= the score
 * of its current K-th best (ties go to the smaller docId, which is already in).  hits, their order, nhits and found are
 * bit-identical to the exhaustive path (tests run both ways).  Multi-term groups are not pruned: their `found` needs the
 * lists merged.  byte_off/counts/idfs as in ns_term_ref; a list given again with another idf is rebuilt. */
int ns_segment_build_blockmax(ns_ctx* ctx, ns_seg* seg, const uint64_t* byte_off, const uint32_t* counts,
                              const float* idfs, uint32_t n_lists);
int ns_ctx_use_pruning(ns_ctx* ctx, int on);
/* on = 0: term groups of exactly two lists take the driver-stream body (table + probes) like every other group instead of
 * the two-list merge body (default: on = 1; both sorted lists advance in lockstep, B's postings find their docs among A's
 * round by a lower bound in LDS: src/api_engine.cpp:449-481 for two lists without a hash table).  Same results. */
int ns_ctx_use_merge(ns_ctx* ctx, int on);
/* Shared term scores.  The reference evaluates the BM25 term score of a posting, src/api_engine.cpp:477-479, once per query
 * that names the posting's list (src/api_engine.cpp:449,464-481: every request walks its lists alone).  The score depends on
 * the list and on the list's idf, not on the query, and a BATCH names the same lists again and again (16384 queries of
 * BASELINE's cfg5 law: ~50 000 term refs, ~40 000 distinct lists, the 32 most frequent ~460 times each).  A sharing batch
 * computes the scores of each DISTINCT list it names once per run — a kernel in front of its scoring kernel, inside
 * ns_batch_run, reading the uploaded {docId, tf} postings and norms and writing {docId, score} (8 B of HBM per posting of
 * the segment, allocated by the first sharing batch) — and its scoring bodies read those.  Nothing is carried from one
 * batch to the next and nothing outlives the run: this is common-subexpression elimination inside one batch, with the
 * reference's operations in the reference's order, and hits, order, nhits, found and score bits are identical (tests run
 * both ways).  mode 1 (default): a batch shares when it scans >= 4 Mi postings and names each distinct posting >= 48 times on
 * average (measured break-even on MI355X: the extra kernel costs ~6 ps per distinct posting, sharing saves ~0.12 ps per use); mode 2: every batch that can (tests); mode 0: never.  A batch never shares when a list of it overlaps another
 * list ever shared in the segment, when a segment of it carries an optional impact stream that lacks one of its lists, or
 * when a list's idf differs from the one a live sharing batch uses; it then scores every posting in place, as with mode 0.
 * ns_batch_info reports NS_INFO_SHARED, shared_lists and shared_postings; sum_score_kernel_ms covers both kernels. */
int ns_ctx_share_scores(ns_ctx* ctx, int mode);

/* ---- one-shot search (host buffers in, host buffers out) ------------------------------------ */
/* hits_out: Q*K entries, query-major, best first: score desc, then seg_id asc, then doc_id asc
 * (the reference leaves ties unspecified: src/api_engine.cpp:485-492); unused tail entries are
 * {-inf, 0xFFFFFFFF, 0xFFFFFFFF}.  nhits_out[Q], found_out[Q] (src/api_engine.cpp:495,505). */
int ns_search_batch(ns_ctx* ctx, const ns_query_desc* queries, const ns_term_ref* terms,
                    uint32_t n_queries, uint32_t k, ns_hit* hits_out, uint32_t* nhits_out,
                    uint64_t* found_out, uint32_t flags);

/* ---- staged search (descriptors resident in HBM; what bench.py times) ----------------------- */
int  ns_batch_prepare(ns_ctx* ctx, const ns_query_desc* queries, const ns_term_ref* terms,
                      uint32_t n_queries, uint32_t k, uint32_t flags, ns_batch** out);
/* Optional: write results into caller-owned DEVICE buffers (e.g. torch tensors that are then
 * all-gathered by RCCL): d_hits Q*K ns_hit, d_nhits Q u32, d_found Q u64.  NULLs restore the
 * batch's own buffers. */
int  ns_batch_bind_outputs(ns_batch* b, void* d_hits, void* d_nhits, void* d_found);
/* Enqueue one pass of the hot path on the ctx stream (asynchronous).  run_flags:
 *   NS_RUN_TIMED  brackets the kernels with HIP events on that stream (read back through ns_batch_get_info after a
 *                 sync or fetch);
 *   NS_RUN_FETCH  also enqueues, right behind the kernels, the copy of the results into pinned host memory and
 *                 records a completion event, so that a later ns_batch_fetch / ns_batch_destroy waits for THIS batch
 *                 only.  That is what lets batches overlap on one ctx:
 *                     prepare(i+1)   host threads regroup and upload while the device scores batch i
 *                     run(i+1, NS_RUN_FETCH)
 *                     fetch(i)       returns as soon as batch i's results have landed
 *                 At most 8 batches may sit between their run and their fetch.  Not for batches with bound outputs. */
#define NS_RUN_TIMED 1
#define NS_RUN_FETCH 2
int  ns_batch_run(ns_batch* b, int run_flags);
/* The hipStream_t this batch's work is enqueued on (the ctx's stream, or its second one under ns_ctx_set_overlap) —
 * for callers that order their own device work behind the batch (bench.py: the RCCL all-gather of a rank's results). */
void* ns_batch_stream(ns_batch* b);
/* Diagnostic for pipelined loops: device time from the end of `prev`'s last kernel to the start of `next`'s first one (both
 * run with NS_RUN_TIMED, both still alive); negative when they overlapped.  NS_E_STATE while `next` has not started. */
int  ns_batch_gap_ms(ns_batch* prev, ns_batch* next, float* ms);
int  ns_batch_sync(ns_batch* b);
int  ns_batch_fetch(ns_batch* b, ns_hit* hits_out, uint32_t* nhits_out, uint64_t* found_out);
int  ns_batch_get_info(ns_batch* b, ns_batch_info* info);
void ns_batch_destroy(ns_batch* b);

/* Semantic query expansion's similarity search: SemanticIndex::most_similar_to_vec,
 * src/semantic_embedding.cpp:104-145.  ns_sem_upload takes the row-major table of L2-normalised fp32 vectors
 * (SemanticIndex::vecs, include/semantic_embedding.hpp:24).  ns_sem_topk: for each of n_q query vectors (host,
 * n_q x dim) the up-to-topk rows with the largest dot product among rows that are not banned for that query
 * (ban_off[n_q + 1] / ban_rows, may be NULL) and have sim >= min_sim — best first, ties to the smaller row;
 * dot products are accumulated in index order in fp32 (the reference's bits).  rows_out / sims_out:
 * n_q x topk (host), counts_out[q] = entries valid for query q.  topk <= 64. */
typedef struct ns_sem ns_sem;
int ns_sem_upload(ns_ctx* ctx, const float* vecs, uint32_t n_rows, uint32_t dim, ns_sem** out);
int ns_sem_release(ns_ctx* ctx, ns_sem* sem);
int ns_sem_topk(ns_ctx* ctx, ns_sem* sem, const float* qvecs, uint32_t n_q, uint32_t topk, float min_sim,
                const uint32_t* ban_off, const uint32_t* ban_rows, uint32_t* rows_out, float* sims_out,
                uint32_t* counts_out, float* device_ms_out);

/* Segment-sharded multi-GPU (the alternative to query sharding for an index that outgrows one
 * GPU's HBM): rank r holds a subset of the segments and scores ALL queries over it; the fixed-size per-rank rows
 * are all-gathered rank-major (hits [n_ranks][n_queries][k], nhits and found [n_ranks][n_queries]) and joined here
 * into the one global heap of src/api_engine.cpp:434-435,485-492 (score desc, global seg asc, doc asc; found =
 * sum, :495).  d_seg_map[r * seg_map_stride + local seg id] = the segment's position in the full manifest
 * (NULL: ids are already global).  All pointers are device pointers; asynchronous on the ctx stream. */
int ns_merge_rank_rows(ns_ctx* ctx, const void* d_hits, const void* d_nhits, const void* d_found, uint32_t n_ranks,
                       uint32_t n_queries, uint32_t k, const uint32_t* d_seg_map, uint32_t seg_map_stride,
                       void* d_out_hits, void* d_out_nhits, void* d_out_found);

/* Index inversion (the step before the path): replaces the per-term std::vector<Posting> +
 * std::sort of the reference's `lexicon` tool (src/lexicon.cpp:52-128).
 *   doc_term_counts[d]  number of (termId, tf) pairs of document d (forward.bin's per-document `cnt`, :63)
 *   pairs               the u32 {termId, tf} pairs of all documents back to back, in file order (:66-67)
 *   n_terms             size of the term dictionary; pairs with termId >= n_terms are dropped (:69)
 * Outputs (host memory): df_out[n_terms] = postings per term; postings_out (capacity n_pairs * 8 bytes) receives
 * the {docId, tf} lists in termId order, each sorted by docId (equal docIds keep file order) — i.e. the
 * reference's inverted_bNNN.bin files concatenated in barrel order; *kept_out = number of postings written;
 * device_ms_out (optional) = HIP-event time of the device part, copies excluded. */
int ns_invert_forward(ns_ctx* ctx, const uint32_t* doc_term_counts, uint32_t n_docs, const uint32_t* pairs,
                      uint64_t n_pairs, uint32_t n_terms, uint32_t* df_out, void* postings_out, uint64_t* kept_out,
                      float* device_ms_out);
/* The same inversion, with the result KEPT on the device as a segment's posting stream (build -> serve without the
 * postings crossing PCIe twice): `seg` is an upload in progress — ns_segment_upload_begin(ctx, id, n_docs, avgdl, doc_len,
 * n_pairs * 8, &seg) with nothing appended yet; the call inverts the forward pairs (doc_term_counts has the segment's
 * n_docs entries), leaves the {docId, tf} lists in termId order in the segment (shorter than announced by the dropped
 * pairs: *kept_out postings), and ns_segment_upload_end(ctx, seg) then publishes it.  df_out[n_terms] gives the
 * caller the lexicon: list t starts at byte 8 * sum(df_out[0..t)) and holds df_out[t] postings.  postings_out (may be
 * NULL) additionally receives the lists in host memory, e.g. to write the inverted files. */
int ns_segment_upload_inverted(ns_ctx* ctx, ns_seg* seg, const uint32_t* doc_term_counts, const uint32_t* pairs,
                               uint64_t n_pairs, uint32_t n_terms, uint32_t* df_out, void* postings_out,
                               uint64_t* kept_out, float* device_ms_out);

/* ---- tuning knobs (per ctx; 0 = library default) --------------------------------------------- */
/* variant: 0 = the product's one scoring launch, k_uscore — every work item picks the driver-stream body, the doc-tile body
 * or (ns_ctx_use_pruning) the block-max body; term groups of more than 64 terms fall back to the workgroup-tile kernel
 * k_score.  libnextsearch_hip.so accepts variant 0 only.  The forced variants — 12..17 = the driver-stream body as a kernel
 * of its own for every group (other table / foreign-budget sizes), 18..20 = the doc-tile body for every group (512 / 1024 /
 * 2048-doc tiles), 1..4 = k_score for every group (four tile sizes) — are test and sweep infrastructure and exist in
 * libnextsearch_hip_variants.so (`make -C nextsearch-api_amd variants`); 5..11 were retired in round 2 and are rejected by
 * both builds.  min_items: number of work items below which groups are additionally split across doc ranges.
 * split_postings: a (query, segment) group is split into doc ranges of about this much estimated work (variant 0: units of
 * one streamed posting, default 98304 for K <= 32 and 131072 above; forced variants: postings). */
int  ns_set_tuning(ns_ctx* ctx, uint32_t variant, uint32_t min_items, uint32_t split_postings);

#ifdef __cplusplus
}
#endif
#endif /* NEXTSEARCH_HIP_H */

// Device-side descriptors shared by the HIP kernels (ns_kernels.hip) and the C-ABI host code
// (ns_api.hip).  Domain vocabulary follows the reference: segments, posting lists, terms, hits.
#pragma once
#include <stdint.h>

namespace ns {

// Accumulators start at +0.0f, as the reference's do (unordered_map value-initialisation, src/api_engine.cpp:480):
// `0.0f + x` is x for every x except -0.0f, which it turns into +0.0f — the reference never returns a negative zero.
//
// The doc-tile body (and k_score, the workgroup-tile fallback for > 64 terms) additionally has to tell an UNTOUCHED slot from a touched one (`found`, :495, counts a doc the
// first time any term hits it, whatever the contribution's value — a posting with tf == 0 contributes an exact zero).
// Its empty slots therefore hold a bit pattern no sum can take: an all-ones NaN.  The first posting that meets it
// counts the doc and starts from +0.0f.  (fp32 arithmetic on finite or infinite inputs never produces this payload;
// an idf or weight that is itself this NaN is outside what the engine can pass.)
static constexpr uint32_t kTileEmptyBits = 0xFFFFFFFFu;
// Packed posting block (f2): 256 postings, stored as byte/halfword PLANES so that one dword load per lane fetches the
// same field of the lane's postings in all four 64-posting chunks (lane l <-> postings l, 64 + l, 128 + l, 192 + l of the block):
//   dword   0 ..  63   tf plane      byte c = min(tf, 255) of chunk c          (255: take tf from the raw stream)
//   dword  64 .. 127   norm plane A  low half = norm index of chunk 0, high half = chunk 1
//   dword 128 .. 191   norm plane B  chunks 2, 3
//   dword 192 ..       doc plane     code 0: one dword, byte c = docId - base of chunk c        (block spans < 256 docs)
//                                    code 1: two dwords (chunks 0|1, chunks 2|3), 16-bit offsets (block spans < 65536 docs)
//                                    code 2: four dwords (chunk c at dword 192 + 64 c), the docIds themselves
// i.e. 4 B (code 0), 5 B (code 1) or 7 B (code 2) per posting are READ; every block owns kPkStrideDwords dwords so that
// block b sits at b * stride (HBM footprint is not what this format saves; bytes moved per posting are).
static constexpr uint32_t kPkBlock = 256;
static constexpr uint32_t kPkStrideDwords = 512;   // 2 KB
static constexpr uint32_t kPkTf = 0, kPkNormA = 64, kPkNormB = 128, kPkDoc = 192;
// Skip tables: one entry per kSkipDocs docs == the doc-tile body's tile (k_uscore: 2 * HK slots).
static constexpr uint32_t kSkipDocs = 1024;
// Block maxima: one fp32 per kBmxBlock postings of a registered list.
static constexpr uint32_t kBmxBlock = 256;

struct DevSeg {
    const uint2* postings;   // {docId, tf} pairs, all inverted files of the segment back to back
    const float* pnorm;      // per POSTING: norm[docId] (streams next to the posting; no dependent gather)
    const float* norm;       // per doc: k1*((1-b) + b*(doc_len/avgdl))   (src/api_engine.cpp:478)
    const uint2* impacts;    // optional {docId, fp32 bits of the BM25 term score}, index-aligned with `postings` (ns_segment_build_impacts); nullptr if never built
    // optional packed posting stream (ns_segment_build_packed): blocks of 256 postings of the SAME posting
    // index space as `postings` (block b = postings [256 b, 256 b + 256)), see kPk* below; nullptr if never built
    const uint32_t* packed;  // kPkStrideDwords dwords per block
    const uint2* pk_hdr;     // per block: {base docId, doc width code}
    const float* ntab;       // the segment's distinct norms; a posting carries a 16-bit index into it
    const float* pk_scores;  // optional, per block 4 x 64 fp32 term scores (chunk-major): the impact stream in packed form
    // optional skip tables (ns_segment_build_skips; SURVEY §8 f2, block metadata): for a registered list, entry i = index
    // (into `postings`) of its first posting with docId >= i * kSkipDocs, i = 0 .. ceil(n_docs / kSkipDocs); one more entry
    // (= the list's end) so that the entry after next can always be read.  nullptr if never built
    const uint32_t* skips;
    // optional block maxima (ns_segment_build_blockmax; SURVEY §8 f2, block-max scores): for a registered list, entry i = the
    // largest BM25 term score — (idf * (tf * 2.2f)) / (tf + norm), src/api_engine.cpp:477-479, with the list's idf — among
    // the postings [256 i, 256 i + 256) of the LIST (block i starts at the list's first posting + 256 i).  nullptr if never built
    const float* blockmax;
    uint64_t     n_postings;
    uint32_t     n_docs;
    uint32_t     n_tiles;    // ceil(n_docs / tile_docs)
};

// One scored posting list of one (query, segment) term group, in query-term order.
struct DevTerm {
    uint64_t list_off;   // first posting (index into DevSeg::postings)
    uint32_t count;      // LexEntry.count
    float    idf;        // bm25_idf(N, df), computed on the host with glibc logf
    float    weight;     // qweight
    uint32_t seg;
    uint32_t skip;       // 0: no skip table; else 1 + index of the list's first entry in DevSeg::skips
    uint32_t bmx;        // 0: no block maxima (or pruning off); else 1 + index of the list's first entry in DevSeg::blockmax
};

// One DISTINCT posting list of a batch whose term scores are computed once for the whole batch (k_share_scores; ns_api.hip
// "shared term scores"): the list's place in its segment, its idf, and the number of postings of the lists before it in the
// batch's build order.  The array ends with a sentinel whose `before` is the total.
struct DevShare {
    uint32_t first;      // first posting (index into DevSeg::postings / DevSeg::impacts)
    uint32_t count;
    float    idf;
    uint32_t seg;
    uint32_t before;     // postings of the lists in front of this one
};

// Work item == one workgroup of k_score: one (query, segment) term group over a range of doc tiles.
struct DevItem {
    uint64_t bounds_off;   // start of this group's [tile][term] boundary table
    uint32_t query;
    uint32_t seg;
    uint32_t term_begin;   // into DevTerm[]
    uint32_t term_count;
    uint32_t tile_begin;
    uint32_t tile_end;
    uint32_t out_slot;     // row of the (partial or final) result arrays
    uint32_t pad;
};

// Work item of the wave-private kernel k_wscore: one WAVE scores one (query, segment) term group
// (<= 64 terms) over the doc range [doc_lo, doc_hi).
struct DevWItem {
    uint32_t query;
    uint32_t seg;
    uint32_t term_begin;   // into DevTerm[]
    uint32_t term_count;   // 1..64
    uint32_t doc_lo;
    uint32_t doc_hi;
    uint32_t out_slot;
    uint32_t whole;        // bit 0: range covers the whole segment (no start/end searches needed); bit 1: doc-tile body;
                           // bit 2: thin foreign lists; bit 3: idf and norms in the short-division range (ns_div_short);
                           // bit 4: some idf or weight of the group has its sign bit set (a contribution may be -0.0f: the
                           //        driver stream then canonicalises its private scores as the reference's 0.0f + x does)
                           // bit 5: doc-tile body on the skip grid: doc_lo is a multiple of kSkipDocs, tiles are grid cells,
                           //        and the terms with DevTerm::skip != 0 take their postings of a tile from the skip table
                           // bit 6: driver-stream bodies: doc_lo (and doc_hi, unless it is n_docs) are multiples of kSkipDocs, and the
                           //        terms with DevTerm::skip != 0 take the range's ends in their lists from the skip table
                           // bit 8: two-list MERGE body (ns_merge_kernel.hip): the group has exactly two term refs; no table
                           // bit 7: PRUNED single-term item (ns_ctx_use_pruning): the group's one list has block maxima (DevTerm::bmx);
                           //        blocks whose best possible score cannot enter the item's top-K are not read; `found` is the
                           //        number of the list's postings in the doc range (one list: every posting is a doc of its own)
};

// Term group == the (query, segment) unit the boundary prepass works on.
struct DevGroup {
    uint64_t bounds_off;
    uint32_t term_begin;
    uint32_t term_count;
    uint32_t seg;
    uint32_t pad;
};

struct DevQuery {
    uint32_t part_begin;   // first partial-result row of this query
    uint32_t part_count;   // number of partial rows (work items); 0 => no scored terms
};

struct Hit {   // == ns_hit
    float    score;
    uint32_t seg;
    uint32_t doc;
};

}  // namespace ns

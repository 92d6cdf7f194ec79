// k_dscore — "driver stream + foreign table" scoring kernel (wave-private, like k_wscore).
//
// One WAVE scores one work item = one (query, segment) term group (<= 64 terms) over a doc range.
// The term with the most postings in the range is the item's DRIVER; all other terms are FOREIGN.
// The wave advances in super-batches [lo, hi]:
//   1. plan windows over the FOREIGN terms only (<= FB postings in total, proportional to what is
//      left of each list), probe each window's last docId: hi = min of those (or the end of the range)
//   2. load the foreign postings (flat, coalesced), BM25 term scores, claim an entry per distinct docId
//      in a BUCKETED table (NB buckets of 4 entries = one ds_read_b128; bucket = low docId bits; entry =
//      (docId - lo) << 8 | index of the posting that owns the doc's accumulator; no LDS atomics: store /
//      read back), accumulate the foreign terms that come BEFORE the driver in query order
//      (read-add-write per term on the owner's accumulator)
//   3. stream the driver's postings with docId <= hi in rounds of 256: BM25 term score, ONE bucket
//      read per posting; a miss (the common case) means no other term has that doc: the doc's score
//      is 0.0f + w*s == w*s exactly, it never touches the table; a hit joins the table accumulation
//   4. accumulate the foreign terms that come AFTER the driver
//   5. owners read the final scores back, count `found`, offer candidates, reset their slots
// Accumulators start at +0.0f like the reference's; `found` counts owners and private postings, never values.
// fp32 accumulation order per doc == query-term order (src/api_engine.cpp:449,480) by construction.
// A hot list joined with sparse lists therefore runs at streaming speed, with the table touched only
// by the sparse postings; dense + dense queries degrade gracefully to the table path for the
// non-driver lists.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ns_internal.h"
#include "ns_wave_kernel.hip"

namespace ns {

#ifdef NS_COUNT
// Diagnostic build only (make -C nextsearch-api_amd count): event counts of the driver-stream body, summed over all
// items of all launches since the last reset; read through ns_debug_counters (tools/dbg/count_run.py).
__device__ unsigned long long g_ns_cnt[20];
#define NS_CNT(i, v) cnt_[(i)] += (unsigned long long)(v)
#else
#define NS_CNT(i, v)
#endif

// NB  buckets of 4 entries per wave   FB  foreign postings per super-batch (<= 256: the owner index has 8 bits).
// A driver lookup reads ONE bucket and stops unless the bucket is full and holds no match: with FB/NB <= 0.75
// a full bucket is a < 1% event, so practically every lookup is a single ds_read_b128 for all 64 lanes.
// One round of the PACKED posting stream (ns_internal.h kPk*): block `blk` of the segment, decoded into the same
// registers a raw round fills — docIds, tf as float (or the precomputed score bits when IMP), norms.
//   PKM == 1  norms come from the per-posting fp32 norm stream (DevSeg::pnorm, chunk-major like a block): 6-7 B per posting
//   PKM == 2  norms come from the block's 16-bit norm index through the table of distinct norms: 4-5 B per posting, but
//             a dependent gather
// Every load of the round is issued before the block header (a scalar load) is looked at: only blocks that span >= 65536
// docs (code 2: tail lists, list boundaries) need two more, dependent loads.
template <bool IMP, int PKM>
__device__ __forceinline__ void pk_decode_round(const DevSeg& seg, const uint32_t blk, const int lane, uint32_t (&doc)[4], float (&tf)[4],
                                                float (&nr)[4], uint32_t (&sbits)[4]) {
    const gp_u32 pb = (gp_u32)seg.packed + (size_t)blk * kPkStrideDwords + (uint32_t)lane;
    const uint32_t d0 = pb[kPkDoc], d1 = pb[kPkDoc + 64];
    uint32_t t = 0, na = 0, nb = 0;
    if (IMP) {
        const gp_f32 sc = (gp_f32)seg.pk_scores + (size_t)blk * kPkBlock + (uint32_t)lane;
#pragma unroll
        for (int c = 0; c < 4; c++) { sbits[c] = __float_as_uint(sc[c * 64]); tf[c] = 0.0f; nr[c] = 0.0f; }
    } else {
        t = pb[kPkTf];
        if (PKM == 2) { na = pb[kPkNormA]; nb = pb[kPkNormB]; }
        else {
            const gp_f32 np = (gp_f32)seg.pnorm + (size_t)blk * kPkBlock + (uint32_t)lane;
#pragma unroll
            for (int c = 0; c < 4; c++) nr[c] = np[c * 64];
        }
    }
    // the header through the constant address space: a wave-uniform address there is a SCALAR load (s_load_dwordx2), counted
    // by lgkmcnt — it does not hold up the vector loads above, which a generic (flat) load followed by readfirstlane did
    typedef const __attribute__((address_space(4))) uint32_t* cp_u32;
    const cp_u32 hp = (cp_u32)(const void*)(seg.pk_hdr + blk);
    const uint32_t base = hp[0];
    const uint32_t code = hp[1];
    if (code == 0u) {
#pragma unroll
        for (int c = 0; c < 4; c++) doc[c] = base + ((d0 >> (8 * c)) & 255u);
    } else if (code == 1u) {
        doc[0] = base + (d0 & 0xFFFFu); doc[1] = base + (d0 >> 16);
        doc[2] = base + (d1 & 0xFFFFu); doc[3] = base + (d1 >> 16);
    } else {
        doc[0] = d0; doc[1] = d1;
        doc[2] = pb[kPkDoc + 128]; doc[3] = pb[kPkDoc + 192];
    }
    if (!IMP) {
        uint32_t tfi[4];
#pragma unroll
        for (int c = 0; c < 4; c++) tfi[c] = (t >> (8 * c)) & 255u;
        // tf >= 255 is stored as the escape 255: the true count is in the raw stream (rare: one test per round)
        const uint32_t nt_ = ~t;
        if (wballot(((nt_ - 0x01010101u) & t & 0x80808080u) != 0u) != 0ull) {   // some byte of t is 0xFF  <=>  some byte of ~t is zero
            const gp_u2 raw = (gp_u2)seg.postings + (size_t)blk * kPkBlock + (uint32_t)lane;
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (tfi[c] == 255u) { const nat_u2 pv = raw[c * 64]; tfi[c] = pv.y; }
        }
        if (PKM == 2) {
            const gp_f32 ntab = (gp_f32)seg.ntab;
            nr[0] = ntab[na & 0xFFFFu]; nr[1] = ntab[na >> 16];
            nr[2] = ntab[nb & 0xFFFFu]; nr[3] = ntab[nb >> 16];
        }
#pragma unroll
        for (int c = 0; c < 4; c++) { tf[c] = (float)tfi[c]; sbits[c] = 0u; }
    }
}

// lanes [lo, hi) of a 64-lane chunk as a mask (lo <= hi <= 64), scalar arithmetic
__device__ __forceinline__ uint64_t lane_span(uint32_t lo, uint32_t hi) {
    const uint64_t upto_hi = hi >= 64u ? ~0ull : ((1ull << hi) - 1ull);
    const uint64_t upto_lo = lo >= 64u ? ~0ull : ((1ull << lo) - 1ull);
    return upto_hi & ~upto_lo;
}

template <int NB, int FB, bool AND, int CB = 256, bool IMP = false, int PK = 0>
__device__ __forceinline__ void dscore_body(const DevWItem& it, const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                            uint32_t* ent, float* vals, uint8_t* mcnt, uint64_t* cand, uint4* tab, uint32_t* aux,
                                            Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                            uint64_t* __restrict__ out_found, uint32_t K, const int lane) {
    // CB: candidate buffer entries, a power of two >= K + 64 (the launcher picks 128 for K <= 64: less LDS)
    constexpr int FE = FB / 64;            // foreign postings per lane per super-batch
    constexpr int DE = 4;                  // driver postings per lane per round
    // entry = (0x8000 | tag) << 16 | home bucket << 8 | owner index; tag = the 15 docId bits above the bucket
    // bits; 0 = empty.  Tag + home bucket pin the low LOG2NB + 15 docId bits, wherever the entry ended up
    // (a full bucket spills into the next one): exact as long as a super-batch spans fewer docs than that.
    constexpr uint32_t EMPTY = 0u;
    constexpr int LOG2NB = (NB == 64) ? 6 : (NB == 128 ? 7 : 8);
    constexpr uint32_t MAXSPAN = (1u << (LOG2NB + 15)) - 1u;   // docs per super-batch - 1
    // Buckets are ORDER-PRESERVING within a super-batch [lo, hi]: bucket(doc) = ((doc - lo) * bm) >> 16 with
    // bm = floor(NB * 2^16 / (hi - lo + 1)), i.e. the super-batch's doc span cut into NB equal ranges (a 24-bit multiply and
    // a bit-field extract: the extract also keeps the index inside the table for docIds outside the span, which are never
    // used).  The docs of one bucket differ by less than 2^15 (bm >= 2 because a super-batch spans at most NB * 2^15 docs),
    // so the low 15 bits of doc - lo plus the home bucket pin the docId exactly.  What the order buys: the postings of ONE
    // term arrive sorted by docId, so the lanes of a term that fall into the same bucket are NEIGHBOURS, and a lane's
    // position inside its bucket is its distance from the first of them — ballot + prefix-max, no claim loop (below).
#define NS_BUCKET(doc) ((uint32_t)__builtin_amdgcn_ubfe(__umul24((doc) - lo, bm), 16u, (uint32_t)LOG2NB))
#define NS_TAG(doc) (0x8000u | (((doc) - lo) & 0x7FFFu))
#define NS_IDENT(doc, bkt) ((NS_TAG(doc) << 16) | ((bkt) << 8))
    static_assert(NB == 64 || NB == 128 || NB == 256, "NB must be 64, 128 or 256");
    static_assert(FB % 64 == 0 && FB >= 64 && FB <= 256 && FB <= 2 * NB, "FB must be a multiple of 64, at most 256 and 2*NB");
    uint4* ent4 = reinterpret_cast<uint4*>(ent);

#ifdef NS_COUNT
    const unsigned long long cyc_t0_ = __builtin_readcyclecounter();
#endif
    const DevSeg seg = segs[it.seg];
    const uint32_t T = it.term_count;
    const bool fast_div = (__builtin_amdgcn_readfirstlane((int)it.whole) & 8) != 0;   // host: every idf and norm of this item is in the range where v_div_scale/v_div_fixup are the identity
    const bool signed_in = (__builtin_amdgcn_readfirstlane((int)it.whole) & 16) != 0;   // host: some idf or weight has its sign bit set
    const gp_u2 postings = (gp_u2)seg.postings;
    const gp_f32 pnorm = (gp_f32)seg.pnorm;
    // IMP: every list of this item has its term scores precomputed (same arithmetic, done once per list at
    // ns_segment_build_impacts): the streams carry {docId, score bits}; no tf, no norm, no division here.
    const gp_u2 stream = IMP ? (gp_u2)seg.impacts : postings;

    {
        const uint4 empty4 = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
#pragma unroll
        for (int g = 0; g < NB / 64; g++) ent4[g * 64 + lane] = empty4;
#pragma unroll
        for (int g = 0; g < FB / 64; g++) {
            vals[g * 64 + lane] = 0.0f;   // the reference's accumulators start at +0.0f (src/api_engine.cpp:480)
            if (AND) mcnt[g * 64 + lane] = 0;
        }
    }

    // ---- lane t owns term t (posting indices are 32-bit: upload rejects segments of >= 2^32 postings) ----
    uint32_t base = 0, cur = 0, end = 0, idf_bits = 0, wq_bits = 0;
    if ((uint32_t)lane < T) {
        const DevTerm tm = terms[it.term_begin + lane];
        base = (uint32_t)tm.list_off;
        idf_bits = __float_as_uint(tm.idf);
        wq_bits = __float_as_uint(tm.weight);
        end = tm.count;
        if (!(it.whole & 1u)) {
            if ((it.whole & 64u) && tm.skip != 0u) {
                // the range's ends come from the list's skip table (the host put doc_lo, and doc_hi unless it is n_docs, on the grid)
                const gp_u32 sk = (gp_u32)seg.skips + (tm.skip - 1u);
                cur = sk[it.doc_lo / kSkipDocs] - base;
                end = sk[(it.doc_hi + (kSkipDocs - 1u)) / kSkipDocs] - base;
            } else {
                const uint2* lst = seg.postings + tm.list_off;
                list_range(lst, tm.count, it.doc_lo, it.doc_hi, seg.n_docs, cur, end);
            }
            if (end < cur) end = cur;
        }
        cur += base;   // absolute posting indices from here on
        end += base;
        tab[lane] = make_uint4(idf_bits, wq_bits, 0u, 0u);   // .z/.w: this super-batch's window, written below
    }
    // ---- the driver: the term with the most postings in this item's range (fixed for the item) ----
    uint32_t dl;
    {
        const uint32_t remv = end - cur;
        const uint32_t mx = wave_max_dpp(remv);
        dl = (uint32_t)__builtin_ctzll(wballot(remv == mx && (uint32_t)lane < T) | (1ull << 63));
        if (dl >= T) dl = 0;
    }
    uint32_t d_cur = rdlane(cur, dl);
    const uint32_t d_end = rdlane(end, dl);
    const float d_idf = __uint_as_float(rdlane(idf_bits, dl));
    const float d_wq = __uint_as_float(rdlane(wq_bits, dl));
    // foreign postings still to be consumed (scalar, saturating)
    uint32_t Rf = 0;
    {
        const uint32_t r32 = ((uint32_t)lane == dl) ? 0u : (end - cur);
        for (uint32_t t = 0; t < T; t++) {
            const uint32_t v = rdlane(r32, t);
            Rf = (Rf + v < Rf) ? 0xFFFFFFFFu : (Rf + v);
        }
    }
    if ((uint32_t)lane == dl) { cur = end; }   // the driver is streamed through d_cur, not through its lane

#ifdef NS_COUNT
    unsigned long long cnt_[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    cnt_[16] = __builtin_readcyclecounter() - cyc_t0_; cnt_[17] = 0;
    NS_CNT(0, 1);
    NS_CNT(10, T);
#endif
    uint32_t lo = it.doc_lo;
    const uint32_t last_doc = it.doc_hi - 1;   // host guarantees doc_hi > doc_lo and doc_hi <= n_docs
    float theta = -__builtin_inff();
    uint32_t ncand = 0;
    uint32_t nsorted = 0;   // leading candidates already in descending order (left by the last shrink)
    uint32_t found_s = 0;      // `found`, wave-uniform: private driver postings and owners of table entries, counted by ballots
    bool ge_mode = false;   // a shrink happened inside the current super-batch: ties with theta may still win on docId

    // offer (score, doc) of the lanes where `cond` holds to the candidate buffer
#define NS_OFFER(cond, scorev, docv)                                                               \
    {                                                                                              \
        bool qf_ = (cond) && ((scorev) > theta);                                                   \
        if (ge_mode) qf_ = (cond) && ((scorev) >= theta);   /* rare: after a shrink inside this super-batch */ \
        unsigned long long mask_ = wballot(qf_);                                                   \
        if (mask_ != 0ull) {                                                                       \
            uint32_t n_ = (uint32_t)__popcll(mask_);                                               \
            if (ncand + n_ > (uint32_t)CB) {                                                       \
                ncand = wave_shrink_cb<CB>(cand, ncand, nsorted, theta, K, lane);                                  \
                ge_mode = true;                                                                    \
                qf_ = (cond) && ((scorev) >= theta);                                               \
                mask_ = wballot(qf_);                                                              \
                n_ = (uint32_t)__popcll(mask_);                                                    \
            }                                                                                      \
            if (qf_) cand[ncand + lanes_below(mask_)] = make_key((scorev), (docv));                \
            ncand += n_;                                                                           \
        }                                                                                          \
    }

    // the same with the condition given as a lane mask in SGPRs
#define NS_OFFER_M(condm, scorev, docv)                                                            \
    {                                                                                              \
        uint64_t mask_ = wballot((scorev) > theta);                                                \
        if (ge_mode) mask_ = wballot((scorev) >= theta);   /* rare: only after a shrink inside this super-batch */ \
        mask_ &= (condm);                                                                          \
        if (mask_ != 0ull) {                                                                       \
            uint32_t n_ = (uint32_t)__popcll(mask_);                                               \
            if (ncand + n_ > (uint32_t)CB) {                                                       \
                ncand = wave_shrink_cb<CB>(cand, ncand, nsorted, theta, K, lane);                                  \
                ge_mode = true;                                                                    \
                mask_ = (condm) & wballot((scorev) >= theta);                                      \
                n_ = (uint32_t)__popcll(mask_);                                                    \
            }                                                                                      \
            if (__builtin_amdgcn_inverse_ballot_w64(mask_)) cand[ncand + lanes_below(mask_)] = make_key((scorev), (docv)); \
            ncand += n_;                                                                           \
        }                                                                                          \
    }

    // Foreign windows: sizes proportional to what is left of each foreign list (all windows span about
    // the same doc range), at most FB postings in total; one docId probe per window (its last posting).
    // Planned right after the cursors move, so the probes fly while the driver streams.
    uint32_t w_n = 0, e_n = 0xFFFFFFFFu;
#define NS_PLAN_FOREIGN()                                                                          \
    {                                                                                              \
        const uint32_t rem_ = end - cur;   /* 0 for the driver's lane and lanes >= T */            \
        const uint32_t nact_ = (uint32_t)__popcll(wballot(rem_ > 0));                             \
        const float scale_ = (float)(FB - (int)nact_) * __builtin_amdgcn_rcpf((float)max(Rf, 1u)); \
        uint32_t w_ = 1u + (uint32_t)((float)rem_ * scale_);                                       \
        w_ = (w_ < rem_) ? w_ : rem_;                                                              \
        const bool probe_ = w_ < rem_;                                                             \
        const uint32_t pi_ = probe_ ? (cur + w_ - 1u) : 0u;                                        \
        const nat_u2 pv_ = postings[pi_];   /* unconditional load of a valid index: no branch */   \
        w_n = (rem_ > 0) ? w_ : 0u;                                                                \
        e_n = probe_ ? pv_.x : 0xFFFFFFFFu;                                                        \
    }
    NS_PLAN_FOREIGN();
    wave_sync();
    for (;;) {
        ge_mode = false;
        NS_CNT(1, 1);
        // ================= 1. foreign windows of this super-batch (planned one super-batch ahead) ==========
        const uint32_t w = w_n;
        const uint32_t e = e_n;
        const uint32_t incl = wave_incl_scan_dpp(w);
        const uint32_t total = rdlane(incl, 63);
        uint32_t hi = wave_min_dpp(e);   // every foreign posting with docId <= hi is inside its window
        // a super-batch spans at most 2^24 - 1 docs (the entry keeps docId - lo in 24 bits)
        const uint32_t span_hi = (last_doc - lo > MAXSPAN) ? (lo + MAXSPAN) : last_doc;
        const bool span_clamped = hi > span_hi && span_hi != last_doc;
        hi = min(hi, span_hi);
        // the bucket function of this super-batch (see NS_BUCKET): bm = floor(NB * 2^16 / (hi - lo + 1)), exact
        uint32_t bm;
        {
            const uint32_t spn1 = (hi >= lo ? hi - lo : 0u) + 1u;   // <= NB * 2^15 (span_hi)
            bm = (uint32_t)((float)((uint32_t)NB << 16) * __builtin_amdgcn_rcpf((float)spn1));   // within a few units of the quotient
            bm = (uint32_t)__builtin_amdgcn_readfirstlane((int)bm);
            while ((uint64_t)bm * spn1 > (uint64_t)((uint32_t)NB << 16)) bm--;
            while ((uint64_t)(bm + 1u) * spn1 <= (uint64_t)((uint32_t)NB << 16)) bm++;   // >= 2, since spn1 <= NB * 2^15
            bm = min(bm, 0xFFFFFFu);   // 24-bit multiplier (spn1 == 1 with NB == 256; the only doc then has offset 0)
        }

        // ================= 2. foreign postings -> table =================
        uint32_t ftj[FE];     // term of the lane's j-th foreign posting; after the claim: term | owner's posting number << 6 | entry index << 14
        float fx[FE];
        bool fok[FE], fmine[FE];
        uint32_t fdoc[FE];
        uint32_t tb_min = 0, tb_max = 0;
#pragma unroll
        for (int j = 0; j < FE; j++) { ftj[j] = 0; fx[j] = 0.0f; fok[j] = false; fmine[j] = false; fdoc[j] = 0; }
        if (total > 0) {
            if ((uint32_t)lane < T) reinterpret_cast<uint2*>(tab + lane)[1] = make_uint2(cur - (incl - w), cur);
            if (T > 8 && (uint32_t)lane < T) aux[lane] = incl;
            wave_sync();
            if (T <= 8) {
                for (uint32_t t = 0; t + 1 < T; t++) {
                    const uint32_t sp = rdlane(incl, t);
#pragma unroll
                    for (int j = 0; j < FE; j++) ftj[j] += ((uint32_t)(j * 64 + lane) >= sp) ? 1u : 0u;
                }
            } else {
#pragma unroll
                for (int j = 0; j < FE; j++) {
                    const uint32_t p = min((uint32_t)(j * 64 + lane), total - 1u);
                    uint32_t a = 0, b = T - 1;   // smallest t with incl[t] > p
                    while (a < b) {
                        const uint32_t m = (a + b) >> 1;
                        if (aux[m] > p) b = m; else a = m + 1;
                    }
                    ftj[j] = a;
                }
                wave_sync();   // aux is reused below
            }
            nat_u2 pst[FE];
            float nrm[FE];
            uint32_t pidx[FE];
#pragma unroll
            for (int j = 0; j < FE; j++) {
                const uint32_t p = (uint32_t)(j * 64 + lane);
                const bool inb = p < total;
                ftj[j] = inb ? ftj[j] : 0u;
                pidx[j] = tab[ftj[j]].z + (inb ? p : 0u);
                pst[j] = stream[pidx[j]];
                nrm[j] = IMP ? 0.0f : pnorm[pidx[j]];
                pst[j].x = inb ? pst[j].x : 0xFFFFFFFFu;   // docId ~0 is never <= hi
            }
            // ---- cursors: the first NOT-taken posting of a window publishes the new cursor ----
            uint32_t batch_consumed;
            {
                if ((uint32_t)lane < T) aux[lane] = cur + w;   // default: whole window consumed
                wave_sync();
                unsigned long long prev_last = 1ull;
#pragma unroll
                for (int j = 0; j < FE; j++) {
                    const bool take = pst[j].x <= hi;
                    fok[j] = take && (pst[j].x >= lo);   // docId < lo only for corrupt (unsorted) lists: consumed, not scored
                    fdoc[j] = pst[j].x;
                    const unsigned long long m = wballot(take);
                    const bool prev_take = (((m << 1) | prev_last) >> lane) & 1ull;
                    prev_last = m >> 63;
                    const bool first_untaken = ((uint32_t)(j * 64 + lane) < total) && !take &&
                                               (prev_take || pidx[j] == tab[ftj[j]].w);
                    if (first_untaken) aux[ftj[j]] = pidx[j];
                }
                wave_sync();
                uint32_t c = 0;
                if ((uint32_t)lane < T && (uint32_t)lane != dl) {
                    const uint32_t ncur = aux[lane];
                    c = ncur - cur;
                    cur = ncur;
                }
                c += dpp_mov<0x111, 0xf>(0u, c);
                c += dpp_mov<0x112, 0xf>(0u, c);
                c += dpp_mov<0x114, 0xf>(0u, c);
                c += dpp_mov<0x118, 0xf>(0u, c);
                c += dpp_mov<0x142, 0xa>(0u, c);
                c += dpp_mov<0x143, 0xc>(0u, c);
                batch_consumed = rdlane(c, 63);
            }
            if (batch_consumed == 0 && !span_clamped) {   // only with corrupt lists (docIds beyond the range): skip the windows
                if ((uint32_t)lane != dl) { cur += w; if (cur > end) cur = end; }
                batch_consumed = total;
            }
            NS_CNT(2, 1);                       // super-batches with foreign postings
            NS_CNT(3, total);                   // foreign postings loaded (window sizes)
            NS_CNT(4, batch_consumed);          // ... of which consumed
            NS_CNT(5, (total + 63) / 64);       // foreign chunks
            NS_CNT(11, (uint32_t)__popcll(wballot(w > 0)));   // active foreign terms
            Rf = (Rf > batch_consumed) ? (Rf - batch_consumed) : 0;
            NS_PLAN_FOREIGN();   // cursors are final: the next super-batch's probes fly from here on
            // ---- BM25 term scores (src/api_engine.cpp:477-480, operation for operation) ----
            tb_min = rdlane(ftj[0], 0);
            {
                const uint32_t lastp = total - 1u;
                uint32_t tl = 0;
#pragma unroll
                for (int j = 0; j < FE; j++)
                    if ((lastp >> 6) == (uint32_t)j) tl = rdlane(ftj[j], lastp & 63u);   // uniform
                tb_max = tl;
            }
            {
                float num[FE], den[FE], wqv[FE];
#pragma unroll
                for (int j = 0; j < FE; j++) {
                    const uint4 te = tab[ftj[j]];
                    const float tf = (float)pst[j].y;
                    den[j] = tf + nrm[j];
                    num[j] = __uint_as_float(te.x) * (tf * (1.2f + 1.0f));
                    wqv[j] = __uint_as_float(te.y);
                }
                if (IMP) {
#pragma unroll
                    for (int j = 0; j < FE; j++) fx[j] = __uint_as_float(pst[j].y);
                } else {
                    ns_div_n<FE>(fx, num, den, fast_div);
                }
#pragma unroll
                for (int j = 0; j < FE; j++) fx[j] = wqv[j] * fx[j];
            }
            // ---- the table of this super-batch: one entry per distinct foreign docId, WITHOUT LDS atomics (integer and
            //      float LDS atomics are serialised per lane on gfx950) ----
            // Pass A, the PRIMARY foreign term (the one with the largest window): its postings are sorted by docId and the
            // bucket function is monotone, so the lanes of one bucket are consecutive lanes; the table is empty when the
            // pass starts, so a lane's position in its bucket is (what the term's previous chunk left there) + (distance
            // from the first lane of the bucket's run) — a ballot of run heads and a prefix maximum of their lane numbers
            // (six DPP steps).  Every such lane stores its entry exactly once: no bucket read, no read-back, no retry.
            // Lanes whose bucket is full (position >= 4) and every other term go through the claim loop of pass B.
            uint32_t pterm = 0xFFFFFFFFu;
            {
                const uint32_t wmax = wave_max_dpp(w);
                if (wmax >= 8u) pterm = (uint32_t)__builtin_ctzll(wballot(w == wmax));
            }
            if (pterm != 0xFFFFFFFFu) {
                uint32_t carry_b = 0xFFFFFFFFu, carry_n = 0u;   // the bucket the term's previous chunk ended in, and its fill
                bool sorted_ok = true;                           // wave-uniform
#pragma unroll
                for (int j = 0; j < FE; j++) {
                    const bool isp = fok[j] && ftj[j] == pterm;
                    const uint64_t pm = wballot(isp);
                    if ((uint32_t)(j * 64) < total && pm != 0ull && sorted_ok) {   // uniform
                        const uint32_t bkt = NS_BUCKET(fdoc[j]);
                        const uint32_t bprev = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - 1) << 2, (int)bkt);
                        const bool prevp = __builtin_amdgcn_inverse_ballot_w64(pm << 1);
                        // sorted docIds give non-decreasing buckets; a list that is not sorted keeps to the claim loop
                        sorted_ok = wballot(isp && prevp && bkt < bprev) == 0ull &&
                                    (carry_b == 0xFFFFFFFFu || rdlane(bkt, (uint32_t)__builtin_ctzll(pm)) >= carry_b);
                        if (sorted_ok) {
                            uint32_t hl = (isp && (!prevp || bkt != bprev)) ? (uint32_t)lane : 0u;   // run heads; prefix maximum = lane of the run's first lane
                            hl = max(hl, dpp_mov<0x111, 0xf>(0u, hl));
                            hl = max(hl, dpp_mov<0x112, 0xf>(0u, hl));
                            hl = max(hl, dpp_mov<0x114, 0xf>(0u, hl));
                            hl = max(hl, dpp_mov<0x118, 0xf>(0u, hl));
                            hl = max(hl, dpp_mov<0x142, 0xa>(0u, hl));
                            hl = max(hl, dpp_mov<0x143, 0xc>(0u, hl));
                            const uint32_t pos = ((bkt == carry_b) ? carry_n : 0u) + ((uint32_t)lane - hl);
                            const uint32_t at = bkt * 4u + pos;
                            if (isp && pos < 4u) {
                                ent[at] = NS_IDENT(fdoc[j], bkt) | (uint32_t)(j * 64 + lane);
                                fmine[j] = true;
                                ftj[j] |= ((uint32_t)(j * 64 + lane) << 6) | (at << 14);
                            }
                            const uint32_t last = 63u - (uint32_t)__builtin_clzll(pm);
                            carry_b = rdlane(bkt, last);
                            carry_n = rdlane(pos, last) + 1u;
                            NS_CNT(15, (uint32_t)__popcll(wballot(isp && pos < 4u)));   // entries placed without a claim
                        }
                    }
                }
                wave_sync();
            }
            // Pass B, everybody else: read the bucket; a matching entry names the doc's owner; otherwise store our entry
            // at the first free position and read it back — a wave's LDS operations execute in order, so exactly one
            // of the colliding entries survives; equal docIds agree on the survivor (the OWNER of the doc's
            // accumulator), everybody else retries on the same bucket (or the next one when it is full).
#pragma unroll
            for (int j = 0; j < FE; j++) {
                if ((uint32_t)(j * 64) >= total) continue;   // uniform
                bool pending = fok[j] && !fmine[j];
                if (wballot(pending) == 0ull) continue;      // uniform: the whole chunk was placed in pass A
                const uint32_t me = (uint32_t)(j * 64 + lane);
                uint32_t b = NS_BUCKET(fdoc[j]);
                const uint32_t mine = NS_IDENT(fdoc[j], b) | me;
                const bool todo = pending;
                uint32_t slot = 0, own = 0;
                bool scan = pending;   // lanes that have to read their (new) bucket
                uint32_t pos = 0;
                for (int round = 0; round < 8 * NB; round++) {
                    if (wballot(pending) == 0ull) break;
                    NS_CNT(6, 1);               // claim iterations
                    if (wballot(scan) != 0ull) {
                        if (scan) {
                            const uint4 q = ent4[b];
                            // entry ^ mine is below 256 exactly for the entry of the same docId (at most one), and then it is
                            // owner ^ me; an empty entry (0) gives `mine`, whose top bit is set: one minimum finds the match
                            const uint32_t tmin = min(min(q.x ^ mine, q.y ^ mine), min(q.z ^ mine, q.w ^ mine));
                            pos = (q.x >> 31) + (q.y >> 31) + (q.z >> 31) + (q.w >> 31);   // entries fill a bucket in order
                            if (tmin < 256u) { own = tmin ^ me; pending = false; }
                            scan = false;
                        }
                    }
                    // Every lane that stores into a bucket in this step saw the same fill level, so they all
                    // target the same position and exactly one entry lands: the losers' next free position
                    // is pos + 1, no need to read the bucket again (same docIds move in lockstep and agree).
                    const bool try_store = pending && pos < 4;
                    if (try_store) ent[b * 4 + pos] = mine;
                    wave_sync();
                    if (try_store) {
                        const uint32_t back = ent[b * 4 + pos];
                        if ((back ^ mine) < 256u) { slot = b * 4 + pos; own = back & 255u; pending = false; }
                        else pos++;
                    } else if (pending) {
                        b = (b + 1) & (uint32_t)(NB - 1);   // full bucket without a match
                        scan = true;
                    }
                    wave_sync();
                }
                if (todo) {
                    fmine[j] = own == me;
                    ftj[j] |= (own << 6) | (slot << 14);
                }
            }
        }

        // term-ordered accumulation of foreign terms in [ta, tb] (read-add-write per term; docIds are
        // unique inside a term, so it is race-free)
#define NS_FOREIGN_RMW(ta, tb)                                                                     \
        for (uint32_t tt = (ta); tt <= (tb); tt++) {                                               \
            if (tt == dl) continue;                                                                \
            NS_CNT(7, 1);                                                                          \
            float old_[FE];                                                                        \
            /* every lane reads (lanes of other terms their own posting's slot, a valid address whose value they drop): */ \
            /* no exec juggling around the reads, only around the writes */                        \
            _Pragma("unroll") for (int j = 0; j < FE; j++) {                                       \
                const bool c_ = fok[j] && (ftj[j] & 63u) == tt;                                    \
                old_[j] = vals[c_ ? ((ftj[j] >> 6) & 255u) : (uint32_t)(j * 64 + lane)];           \
            }                                                                                      \
            _Pragma("unroll") for (int j = 0; j < FE; j++) {                                       \
                if (fok[j] && (ftj[j] & 63u) == tt) {                                              \
                    const uint32_t o_ = (ftj[j] >> 6) & 255u;                                      \
                    vals[o_] = old_[j] + fx[j];                                                    \
                    if (AND) mcnt[o_] = (uint8_t)(mcnt[o_] + 1);                                   \
                }                                                                                  \
            }                                                                                      \
            wave_sync();                                                                           \
        }
        if (total > 0 && tb_min < dl) { NS_FOREIGN_RMW(tb_min, min(tb_max, dl - 1)); }

        // ================= 3. stream the driver's postings with docId <= hi =================
        // Lane predicates of this section live in SGPR pairs (ballot masks combined with scalar logic);
        // the loads use a scalar base + a fixed lane offset (they may run up to 255 postings past the
        // end of the list: the device buffers are padded), so the only vector work per posting is the
        // docId compare, the BM25 term score and the bucket probe.
        bool driver_progress = false;
        for (;;) {
            const uint32_t remd = d_end - d_cur;
            if (remd == 0) break;
            uint32_t n;
            nat_u2 ps[DE];
            float nr[DE];
            float tfv[DE];
            uint32_t sbits[DE];
            uint32_t cnt = 0, r_hits = 0;
            NS_CNT(8, 1);                       // driver rounds (256 loaded each)
            float dx[DE];
            uint64_t dokm[DE];   // postings of this round that belong to the super-batch and are still private
            uint32_t first_pos, last_pos;       // PK: positions (0..255) of the round's first / last valid posting inside its block
            if (PK) {
                // the packed stream comes in blocks of 256 postings of the segment's posting index space: a round is the
                // part [a, e) of the block that holds the cursor (the cursor moves by what is consumed, so a block may be
                // decoded again by the next super-batch, as a raw round may be loaded again)
                static_assert(DE == 4, "a packed round is one block of 4 chunks");
                const uint32_t blk = d_cur >> 8;
                const uint32_t a = d_cur & 255u;
                const uint32_t e = min(256u, d_end - (blk << 8));
                n = e - a;
                first_pos = a; last_pos = e - 1u;
                uint32_t docs_[4];
                pk_decode_round<IMP, PK>(seg, blk, lane, docs_, tfv, nr, sbits);
#pragma unroll
                for (int j = 0; j < DE; j++) { ps[j].x = docs_[j]; ps[j].y = sbits[j]; }
                if (n == (uint32_t)(DE * 64)) {   // a whole block (all but the first and the last of a list's range): no lane masks to build
#pragma unroll
                    for (int j = 0; j < DE; j++) {
                        dokm[j] = wballot(ps[j].x <= hi);
                        cnt += (uint32_t)__popcll(dokm[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < DE; j++) {
                        const uint32_t c0 = (uint32_t)(j * 64);
                        const uint64_t vm = lane_span(a > c0 ? a - c0 : 0u, e > c0 ? min(e - c0, 64u) : 0u);
                        dokm[j] = wballot(ps[j].x <= hi) & vm;
                        cnt += (uint32_t)__popcll(dokm[j]);
                    }
                }
            } else {
                n = min(remd, (uint32_t)(DE * 64));
                first_pos = 0u; last_pos = n - 1u;
                const gp_u2 sp = stream + d_cur;
                const gp_f32 np = pnorm + d_cur;
#pragma unroll
                for (int j = 0; j < DE; j++) {
                    ps[j] = sp[j * 64 + lane];
                    nr[j] = IMP ? 0.0f : np[j * 64 + lane];
                    tfv[j] = (float)ps[j].y;
                }
                if (n == (uint32_t)(DE * 64)) {   // a full round (all but a list's last): no lane mask to build
#pragma unroll
                    for (int j = 0; j < DE; j++) {
                        dokm[j] = wballot(ps[j].x <= hi);
                        cnt += (uint32_t)__popcll(dokm[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < DE; j++) {
                        const uint32_t left = (n > (uint32_t)(j * 64)) ? (n - (uint32_t)(j * 64)) : 0u;   // scalar
                        const uint64_t nmask = (left >= 64u) ? ~0ull : ((1ull << left) - 1ull);
                        dokm[j] = wballot(ps[j].x <= hi) & nmask;
                        cnt += (uint32_t)__popcll(dokm[j]);
                    }
                }
            }
            NS_CNT(12, (n + 63) / 64);          // driver chunks with postings
            {   // src/api_engine.cpp:477-480, operation for operation
                float num[DE], den[DE];
#pragma unroll
                for (int j = 0; j < DE; j++) {
                    const float tf = tfv[j];
                    den[j] = tf + nr[j];
                    num[j] = d_idf * (tf * (1.2f + 1.0f));
                }
                if (IMP) {
#pragma unroll
                    for (int j = 0; j < DE; j++) dx[j] = __uint_as_float(ps[j].y);
                } else {
                    ns_div_n<DE>(dx, num, den, fast_div);
                }
#pragma unroll
                for (int j = 0; j < DE; j++) dx[j] = d_wq * dx[j];
                // A private posting's doc score is the reference's 0.0f + w*s.  That is w*s itself unless w*s is -0.0f,
                // which needs a sign bit in idf or weight (a posting with tf == 0 then gives it): rare enough for a
                // wave-uniform branch around one add per posting.
                if (signed_in) {
#pragma unroll
                    for (int j = 0; j < DE; j++) dx[j] = 0.0f + dx[j];
                }
            }
            // does any foreign doc of this super-batch fall into the doc range of this round?  (usually
            // not when the driver is much denser than the foreign lists: then the lookups are skipped)
            bool any_foreign_here = total > 0;
            if (FB <= 64 && total > 0) {   // only worth testing for the thin-foreign class
                uint32_t rfirst, rlast;
                if (n == (uint32_t)(DE * 64)) {
                    rfirst = rdlane(ps[0].x, 0);
                    rlast = rdlane(ps[DE - 1].x, 63);
                } else {
                    rfirst = 0; rlast = 0;
#pragma unroll
                    for (int j = 0; j < DE; j++) {   // uniform selects
                        if ((first_pos >> 6) == (uint32_t)j) rfirst = rdlane(ps[j].x, first_pos & 63u);
                        if ((last_pos >> 6) == (uint32_t)j) rlast = rdlane(ps[j].x, last_pos & 63u);
                    }
                }
                bool here = false;
#pragma unroll
                for (int j = 0; j < FE; j++) here = here || (fok[j] && fdoc[j] >= rfirst && fdoc[j] <= rlast);
                any_foreign_here = wballot(here) != 0ull;
            }
            if (any_foreign_here) {
                // one bucket read per posting; a hit joins the table accumulation at the driver's place in
                // the term order (foreign terms before it are already in, those after it follow)
#pragma unroll
                for (int j = 0; j < DE; j++) {
                    if (dokm[j] == 0ull) continue;   // uniform: the super-batch takes no posting of this chunk (docIds ascend: the round that reaches `hi` usually leaves its last chunks untaken)
                    const uint32_t tag = NS_TAG(ps[j].x);
                    const uint32_t hb = NS_BUCKET(ps[j].x);   // in range whatever the docId; only postings of this super-batch count (dokm)
                    uint32_t b = hb;
                    uint4 q = ent4[b];
                    // first the 16-bit tags (one compare per entry); a tag match is verified below
                    const uint64_t cm = wballot((q.x >> 16) == tag) | wballot((q.y >> 16) == tag) |
                                        wballot((q.z >> 16) == tag) | wballot((q.w >> 16) == tag);
                    // full bucket without a tag match: the doc may sit in the next bucket (a 0.2% event per lane)
                    const uint64_t go = dokm[j] & (cm | wballot(q.w != EMPTY));
                    if (go != 0ull) {
                        const uint32_t ident = NS_IDENT(ps[j].x, hb);
                        uint32_t m = 0;
                        bool more = __builtin_amdgcn_inverse_ballot_w64(go);
                        while (wballot(more) != 0ull) {
                            if (more) {
                                m = ((q.x ^ ident) < 256u) ? q.x : m;
                                m = ((q.y ^ ident) < 256u) ? q.y : m;
                                m = ((q.z ^ ident) < 256u) ? q.z : m;
                                m = ((q.w ^ ident) < 256u) ? q.w : m;
                                more = m == 0 && q.w != EMPTY;
                                if (more) { b = (b + 1) & (uint32_t)(NB - 1); q = ent4[b]; }
                            }
                        }
                        const uint64_t hitm = wballot(m != 0);   // only lanes of `go` ever set m
                        if (hitm != 0ull) {
                            if (m != 0) {
                                const uint32_t o = m & 255u;
                                vals[o] = vals[o] + dx[j];
                                if (AND) mcnt[o] = (uint8_t)(mcnt[o] + 1);
                            }
                            dokm[j] &= ~hitm;   // scored through the table's owner
                            r_hits += (uint32_t)__popcll(hitm);
                        }
                    }
                }
            }
            // private postings: no other term has the doc: score == 0.0f + w*s == w*s exactly
            if (!AND || T == 1) {   // conjunctive extension: one term alone never qualifies
                found_s += cnt - r_hits;   // every taken posting that did not join a table entry is a doc of its own
#pragma unroll
                for (int j = 0; j < DE; j++) {
                    NS_OFFER_M(dokm[j], dx[j], ps[j].x);
                }
            }
            NS_CNT(9, cnt);                     // driver postings consumed
            NS_CNT(13, any_foreign_here ? (n + 63) / 64 : 0);   // chunks that probed the table
            NS_CNT(14, r_hits);
            d_cur += cnt;
            driver_progress = driver_progress || (cnt > 0);
            if (cnt < n) break;   // reached hi
        }

        // ================= 4. foreign terms after the driver, 5. read back through the owners =================
        if (total > 0) {
            if (tb_max > dl) { NS_FOREIGN_RMW(max(tb_min, dl + 1), tb_max); }
            float fin[FE];
            bool scored[FE];
#pragma unroll
            for (int j = 0; j < FE; j++) {
                // (every lane reads and clears its own posting's slot — a non-owner's is unused and stays +0.0f: no exec
                // juggling around the accumulators, only around the table entries, which only their owners may touch)
                fin[j] = vals[j * 64 + lane];
                scored[j] = fmine[j];
                if (AND && fmine[j]) scored[j] = (mcnt[j * 64 + lane] == (uint8_t)T);   // conjunctive extension: every term ref hit the doc
            }
            wave_sync();
#pragma unroll
            for (int j = 0; j < FE; j++) {
                vals[j * 64 + lane] = 0.0f;
                if (fmine[j]) {   // the owner resets its entry for the next super-batch
                    ent[ftj[j] >> 14] = EMPTY;
                    if (AND) mcnt[j * 64 + lane] = 0;
                }
                found_s += (uint32_t)__popcll(wballot(scored[j]));
                NS_OFFER(scored[j], fin[j], fdoc[j]);
            }
            wave_sync();
        }
        if (ncand > (uint32_t)(CB - 64)) ncand = wave_shrink_cb<CB>(cand, ncand, nsorted, theta, K, lane);   // keep room for one more step of offers

        if (hi >= last_doc) break;
        if (Rf == 0 && d_cur >= d_end) break;
        if (total == 0 && !driver_progress) break;   // only with corrupt lists: nothing can advance
        lo = hi + 1;
    }
#undef NS_FOREIGN_RMW
#undef NS_OFFER
#undef NS_OFFER_M
#undef NS_PLAN_FOREIGN
#undef NS_TAG
#undef NS_IDENT
#undef NS_BUCKET

#ifdef NS_COUNT
    const unsigned long long cyc_t1_ = __builtin_readcyclecounter();
#endif
    // ---- this item's top-K ----
    wave_sync();
    ncand = wave_shrink_cb<CB>(cand, ncand, nsorted, theta, K, lane);
    const uint32_t n = min(ncand, K);
    Hit* oh = out_hits + (uint64_t)it.out_slot * K;
    for (uint32_t i = lane; i < K; i += 64) {
        Hit h;
        if (i < n) {
            const uint64_t key = cand[i];
            h.score = unorder_bits((uint32_t)(key >> 32));
            h.seg = it.seg;
            h.doc = 0xFFFFFFFFu - (uint32_t)key;
        } else {
            h.score = -__builtin_inff();
            h.seg = 0xFFFFFFFFu;
            h.doc = 0xFFFFFFFFu;
        }
        oh[i] = h;
    }
    if (lane == 63) {
        out_nhits[it.out_slot] = n;
        out_found[it.out_slot] = (uint64_t)found_s;
    }
#ifdef NS_COUNT
    cnt_[17] = __builtin_readcyclecounter() - cyc_t0_;
    cnt_[18] = __builtin_readcyclecounter() - cyc_t1_;
    if (lane == 0)
        for (int i = 0; i < 20; i++) if (cnt_[i]) atomicAdd(&g_ns_cnt[i], cnt_[i]);
#endif
}

// HK = table entries per wave (4 per bucket)
template <int HK, int FB, bool AND>
__global__ void __launch_bounds__(256) k_dscore(const DevWItem* __restrict__ items, uint32_t n_items,
                                                const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                                Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                                uint64_t* __restrict__ out_found, uint32_t K) {
    constexpr int WPB = 4;                 // independent waves per workgroup
    __shared__ __attribute__((aligned(16))) uint32_t s_ent[WPB][HK];
    __shared__ __attribute__((aligned(16))) float s_vals[WPB][FB];
    __shared__ __attribute__((aligned(16))) uint8_t s_mcnt[WPB][AND ? FB : 16];   // AND: term refs that hit the doc
    __shared__ uint64_t s_cand[WPB][256];
    __shared__ __attribute__((aligned(16))) uint4 s_tab[WPB][64];   // per term: {idf, qweight, first posting - excl prefix, first posting}
    __shared__ uint32_t s_aux[WPB][64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: the item is fetched with scalar loads
    const int lane = threadIdx.x & 63;
    const uint32_t item_idx = blockIdx.x * WPB + wave;
    if (item_idx >= n_items) return;   // whole wave leaves; there is no workgroup barrier in this kernel
    const DevWItem it = items[item_idx];
    dscore_body<HK / 4, FB, AND>(it, terms, segs, s_ent[wave], s_vals[wave], s_mcnt[wave], s_cand[wave], s_tab[wave], s_aux[wave],
                                 out_hits, out_nhits, out_found, K, lane);
}

}  // namespace ns

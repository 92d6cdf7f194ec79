// mscore_body — two-list MERGE body (north_star: "intersect/merge across query terms with wavefront ballot/prefix-sum
// primitives"; src/api_engine.cpp:449-481 for a group of exactly two scored lists).
//
// The driver-stream body hashes every posting of the non-driver list into an LDS table that every driver posting then
// probes.  For a group of TWO lists neither table nor claims are needed: both lists are sorted by docId, so the wave
// advances through them in lockstep —
//   1. load a round of the longer list A (256 postings, 4 per lane) and a window of the other list B (64 .. 256 postings,
//      sized by what is left of the two lists); hi = the smaller of the two loads' last docIds (a list that ends inside its
//      load sets no bound): every posting <= hi of BOTH lists is in registers, and at least one load is consumed whole;
//   2. BM25 term scores of both (src/api_engine.cpp:477-479, operation for operation);
//   3. A's docIds and scores go to LDS in posting order (ascending; postings beyond hi as ~0); every B lane finds its doc
//      there by an 8-step lower bound (a wave's LDS operations execute in order; the searches of all B chunks finish
//      before anything is marked), takes A's score and marks A's slot; A's lanes read their slots back: marked = matched;
//   4. a doc of both lists scores (0.0f + x_first) + x_second in QUERY-TERM order (:449,:480), a doc of one list 0.0f + x;
//      `found` (:495) = |A| + |B| - |A and B| by popcounts; candidates above theta go to the wave's buffer (:485-492).
// No table, no claims, no probes, no accumulation passes, no read-back.  The conjunctive extension keeps the matched docs only.
// Selected by the host for general-class groups of exactly two term refs (DevWItem::whole bit 8, ns_ctx_use_merge).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ns_internal.h"
#include "ns_wave_kernel.hip"

namespace ns {

// IMP: both lists have their term scores in the segment's score stream ({docId, score bits}, index-aligned with the postings:
// the optional impact stream, or the batch's shared term scores): no tf, no norm, no division here (see dscore_body).
template <bool AND, int CB, bool IMP = false>
__device__ __forceinline__ void mscore_body(const DevWItem& it, const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                            uint32_t* l_docs /* 256 */, float* l_sc /* 256 */, uint64_t* cand,
                                            Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                            uint64_t* __restrict__ out_found, uint32_t K, const int lane) {
    constexpr int DE = 4;    // A: postings per lane per round
    constexpr int BE = 4;    // B: at most this many chunks of 64 per window
    constexpr uint32_t kNone = 0xFFFFFFFFu, kMark = 0xFFFFFFFEu;
    const DevSeg seg = segs[it.seg];
    const bool fast_div = (__builtin_amdgcn_readfirstlane((int)it.whole) & 8) != 0;
    const gp_u2 postings = IMP ? (gp_u2)seg.impacts : (gp_u2)seg.postings;
    const gp_f32 pnorm = (gp_f32)seg.pnorm;

    // ---- lanes 0 and 1 own the two terms: absolute posting range of the item in each list ----
    uint32_t cur = 0, end = 0, idf_bits = 0, wq_bits = 0;
    if (lane < 2) {
        const DevTerm tm = terms[it.term_begin + lane];
        const uint32_t base = (uint32_t)tm.list_off;
        idf_bits = __float_as_uint(tm.idf);
        wq_bits = __float_as_uint(tm.weight);
        end = tm.count;
        if (!(it.whole & 1u)) {
            if ((it.whole & 64u) && tm.skip != 0u) {
                const gp_u32 sk = (gp_u32)seg.skips + (tm.skip - 1u);
                cur = sk[it.doc_lo / kSkipDocs] - base;
                end = sk[(it.doc_hi + (kSkipDocs - 1u)) / kSkipDocs] - base;
            } else {
                list_range(seg.postings + tm.list_off, tm.count, it.doc_lo, it.doc_hi, seg.n_docs, cur, end);
            }
            if (end < cur) end = cur;
        }
        cur += base;
        end += base;
    }
    // A = the list with more postings in the range (ties: the first term); its index decides the accumulation order
    const uint32_t rem0 = rdlane(end, 0) - rdlane(cur, 0), rem1 = rdlane(end, 1) - rdlane(cur, 1);
    const uint32_t ai = rem1 > rem0 ? 1u : 0u, bi = ai ^ 1u;
    uint32_t a_cur = rdlane(cur, ai), b_cur = rdlane(cur, bi);
    const uint32_t a_end = rdlane(end, ai), b_end = rdlane(end, bi);
    const float a_idf = __uint_as_float(rdlane(idf_bits, ai)), b_idf = __uint_as_float(rdlane(idf_bits, bi));
    const float a_wq = __uint_as_float(rdlane(wq_bits, ai)), b_wq = __uint_as_float(rdlane(wq_bits, bi));
    const bool a_first = ai < bi;   // A precedes B in query-term order

    const uint32_t last_doc = it.doc_hi - 1;
    float theta = -__builtin_inff();
    uint32_t ncand = 0, nsorted = 0, found_s = 0;
    bool ge_mode = false;

#define NS_MOFFER(condm, scorev, docv)                                                             \
    {                                                                                              \
        uint64_t mask_ = ge_mode ? wballot((scorev) >= theta) : wballot((scorev) > theta);         \
        mask_ &= (condm);                                                                          \
        if (mask_ != 0ull) {                                                                       \
            uint32_t n_ = (uint32_t)__popcll(mask_);                                               \
            if (ncand + n_ > (uint32_t)CB) {                                                       \
                ncand = wave_shrink_cb<CB>(cand, ncand, nsorted, theta, K, lane);                  \
                ge_mode = true;   /* ties with theta may still win on docId inside this step */    \
                mask_ = (condm) & wballot((scorev) >= theta);                                      \
                n_ = (uint32_t)__popcll(mask_);                                                    \
            }                                                                                      \
            if (__builtin_amdgcn_inverse_ballot_w64(mask_)) cand[ncand + lanes_below(mask_)] = make_key((scorev), (docv)); \
            ncand += n_;                                                                           \
        }                                                                                          \
    }

    wave_sync();
    for (;;) {
        const uint32_t a_rem = a_end - a_cur, b_rem = b_end - b_cur;
        if (a_rem == 0u && b_rem == 0u) break;
        ge_mode = false;
        const uint32_t na = min(a_rem, (uint32_t)(DE * 64));
        // B's window: what B is expected to hold under one round of A (the lists thin out alike), a quarter more, whole chunks
        uint32_t nb;
        {
            const float want = (float)na * ((float)b_rem * __builtin_amdgcn_rcpf((float)max(a_rem, 1u))) * 1.25f + 24.0f;
            const uint32_t w = a_rem == 0u ? (uint32_t)(BE * 64) : (uint32_t)min(want, (float)(BE * 64));
            nb = min(b_rem, min((uint32_t)(BE * 64), (w + 63u) & ~63u));
            nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
        }
        // ---- loads: scalar base + fixed lane offset (the buffers are padded past the last list) ----
        nat_u2 pa[DE], pb[BE];
        float na_[DE], nb_[BE];
        {
            const gp_u2 sa = postings + a_cur;
            const gp_f32 fa = pnorm + a_cur;
#pragma unroll
            for (int j = 0; j < DE; j++) {
                if ((uint32_t)(j * 64) >= na) { pa[j] = nat_u2{kNone, 0u}; na_[j] = 1.0f; continue; }   // uniform
                pa[j] = sa[j * 64 + lane];
                na_[j] = IMP ? 1.0f : fa[j * 64 + lane];
            }
            const gp_u2 sb = postings + b_cur;
            const gp_f32 fb = pnorm + b_cur;
#pragma unroll
            for (int j = 0; j < BE; j++) {
                if ((uint32_t)(j * 64) >= nb) { pb[j] = nat_u2{kNone, 0u}; nb_[j] = 1.0f; continue; }   // uniform
                pb[j] = sb[j * 64 + lane];
                nb_[j] = IMP ? 1.0f : fb[j * 64 + lane];
            }
        }
        // ---- hi: every posting <= hi of both lists is loaded ----
        uint32_t hi = last_doc;
        if (a_rem > na) {   // A goes on beyond this round: its last loaded docId bounds the step (na == 256 here)
            hi = min(hi, rdlane(pa[DE - 1].x, 63));
        }
        if (b_rem > nb) {
            uint32_t lastb = 0;
            const uint32_t lp = nb - 1u;
#pragma unroll
            for (int j = 0; j < BE; j++)
                if ((lp >> 6) == (uint32_t)j) lastb = rdlane(pb[j].x, lp & 63u);   // uniform select
            hi = min(hi, lastb);
        }
        // ---- who is taken (lanes inside the load with docId <= hi) ----
        uint64_t ta[DE], tb[BE];
        uint32_t cnt_a = 0, cnt_b = 0;
#pragma unroll
        for (int j = 0; j < DE; j++) {
            const uint32_t left = (na > (uint32_t)(j * 64)) ? (na - (uint32_t)(j * 64)) : 0u;
            const uint64_t nm = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
            ta[j] = wballot(pa[j].x <= hi) & nm;
            cnt_a += (uint32_t)__popcll(ta[j]);
        }
#pragma unroll
        for (int j = 0; j < BE; j++) {
            const uint32_t left = (nb > (uint32_t)(j * 64)) ? (nb - (uint32_t)(j * 64)) : 0u;
            const uint64_t nm = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
            tb[j] = wballot(pb[j].x <= hi) & nm;
            cnt_b += (uint32_t)__popcll(tb[j]);
        }
        if (cnt_a + cnt_b == 0u) {   // only with lists that are not docId-ascending: consume the loads, score nothing
            a_cur += na; b_cur += nb;
            continue;
        }
        // ---- BM25 term scores (src/api_engine.cpp:477-480) ----
        float xa[DE], xb[BE];
        if (IMP) {
#pragma unroll
            for (int j = 0; j < DE; j++) xa[j] = a_wq * __uint_as_float(pa[j].y);
#pragma unroll
            for (int j = 0; j < BE; j++) xb[j] = b_wq * __uint_as_float(pb[j].y);
        } else {
        {
            float num[DE], den[DE];
#pragma unroll
            for (int j = 0; j < DE; j++) {
                const float tf = (float)pa[j].y;
                den[j] = tf + na_[j];
                num[j] = a_idf * (tf * (1.2f + 1.0f));
            }
            ns_div_n<DE>(xa, num, den, fast_div);
#pragma unroll
            for (int j = 0; j < DE; j++) xa[j] = a_wq * xa[j];
        }
        {
            float num[BE], den[BE];
#pragma unroll
            for (int j = 0; j < BE; j++) {
                const float tf = (float)pb[j].y;
                den[j] = tf + nb_[j];
                num[j] = b_idf * (tf * (1.2f + 1.0f));
            }
            ns_div_n<BE>(xb, num, den, fast_div);
#pragma unroll
            for (int j = 0; j < BE; j++) xb[j] = b_wq * xb[j];
        }
        }
        // ---- A's taken postings to LDS in posting order (ascending docIds; everything else ~0) ----
#pragma unroll
        for (int j = 0; j < DE; j++) {
            l_docs[j * 64 + lane] = __builtin_amdgcn_inverse_ballot_w64(ta[j]) ? pa[j].x : kNone;
            l_sc[j * 64 + lane] = xa[j];
        }
        wave_sync();
        // ---- every B lane: lower bound of its docId among A's 256 slots ----
        uint32_t posb[BE];
        uint64_t mb[BE];   // B lanes whose doc is also in A
        uint32_t n_match = 0;
#pragma unroll
        for (int j = 0; j < BE; j++) {
            posb[j] = 0; mb[j] = 0ull;
            if (tb[j] == 0ull) continue;   // uniform
            uint32_t p = 0;
#pragma unroll
            for (uint32_t st = 128; st > 0; st >>= 1) p += (l_docs[p + st - 1u] < pb[j].x) ? st : 0u;
            posb[j] = p;
            mb[j] = wballot(l_docs[p] == pb[j].x) & tb[j];
            n_match += (uint32_t)__popcll(mb[j]);
        }
        // matched B lanes take A's score and mark A's slot (all searches are done: the marks cannot disturb them)
        float xam[BE];
#pragma unroll
        for (int j = 0; j < BE; j++) {
            xam[j] = 0.0f;
            if (mb[j] == 0ull) continue;   // uniform
            xam[j] = l_sc[posb[j]];   // (every lane reads: posb is a valid slot for all of them; only matched lanes use the value)
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < BE; j++) {
            if (mb[j] == 0ull) continue;   // uniform
            if (__builtin_amdgcn_inverse_ballot_w64(mb[j])) l_docs[posb[j]] = kMark;
        }
        wave_sync();
        // ---- offers ----
        if (!AND) {
            found_s += cnt_a + cnt_b - n_match;
#pragma unroll
            for (int j = 0; j < DE; j++) {
                if (ta[j] == 0ull) continue;   // uniform
                uint64_t priv = ta[j];
                if (n_match != 0u) priv &= ~wballot(l_docs[j * 64 + lane] == kMark);   // scored by B's lane
                const float s = 0.0f + xa[j];   // the reference's accumulators start at +0.0f (:480)
                NS_MOFFER(priv, s, pa[j].x);
            }
        } else {
            found_s += n_match;   // conjunctive extension: both term refs hit the doc
        }
#pragma unroll
        for (int j = 0; j < BE; j++) {
            const uint64_t who = AND ? mb[j] : tb[j];
            if (who == 0ull) continue;   // uniform
            const bool both = __builtin_amdgcn_inverse_ballot_w64(mb[j]);
            const float x1 = both ? (a_first ? xam[j] : xb[j]) : xb[j];
            float s = 0.0f + x1;
            if (both) s = s + (a_first ? xb[j] : xam[j]);
            NS_MOFFER(who, s, pb[j].x);
        }
        wave_sync();   // the next step rewrites the LDS arrays
        a_cur += cnt_a;
        b_cur += cnt_b;
        if (ncand > (uint32_t)(CB - 64)) ncand = wave_shrink_cb<CB>(cand, ncand, nsorted, theta, K, lane);
    }
#undef NS_MOFFER

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
}

}  // namespace ns

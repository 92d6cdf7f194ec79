// k_dscore — "driver stream + foreign table" scoring kernel (wave-private, like k_wscore).
//
// One WAVE scores one work item = one (query, segment) term group (<= 64 terms) over a doc range.
// The term with the most postings in the range is the item's DRIVER; all other terms are FOREIGN.
// The wave advances in super-batches [lo, hi]:
//   1. plan windows over the FOREIGN terms only (<= HK/2 postings in total, proportional to what is
//      left of each list), probe each window's last docId: hi = min of those (or the end of the range)
//   2. load the foreign postings (flat, coalesced), BM25 term scores, claim a slot per distinct docId
//      in an open-addressing table (no LDS atomics: store / read back), elect one owner per slot,
//      accumulate the foreign terms that come BEFORE the driver in query order (read-add-write per term)
//   3. stream the driver's postings with docId <= hi in rounds of 256: BM25 term score, ONE table
//      lookup per posting; a miss (the common case) means no other term has that doc: the doc's score
//      is 0.0f + w*s == w*s exactly, it never touches the table; a hit joins the table accumulation
//   4. accumulate the foreign terms that come AFTER the driver
//   5. owners read the final scores back, count `found`, offer candidates, reset their slots
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

// HK  table slots per wave        FB  foreign postings per super-batch (load factor FB/HK: driver lookups
//                                    are UNSUCCESSFUL searches, whose probe length grows fast with the load)
template <int HK, int FB, bool AND, int CB = 256>
__device__ __forceinline__ void dscore_body(const DevWItem& it, const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                            float* vals, uint32_t* keys, uint8_t* mcnt, uint64_t* cand, uint4* tab, uint32_t* aux,
                                            Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                            uint64_t* __restrict__ out_found, uint32_t K, const int lane) {
    // CB: candidate buffer entries, a power of two >= K + 64 (the launcher picks 128 for K <= 64: less LDS, one more workgroup per CU)
    constexpr int FE = FB / 64;            // foreign postings per lane per super-batch
    constexpr int DE = 4;                  // driver postings per lane per round
    constexpr int LOG2HK = (HK == 256) ? 8 : (HK == 512 ? 9 : 10);
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    static_assert(HK == 256 || HK == 512 || HK == 1024, "HK must be 256, 512 or 1024");
    static_assert(FB % 64 == 0 && FB >= 64 && FB <= HK / 2, "FB must be a multiple of 64, at most HK/2");

    const DevSeg seg = segs[it.seg];
    const uint32_t T = it.term_count;
    const gp_u2 postings = (gp_u2)seg.postings;
    const gp_f32 pnorm = (gp_f32)seg.pnorm;

    {
        const float4 sent4 = make_float4(__uint_as_float(kSentinelBits), __uint_as_float(kSentinelBits),
                                         __uint_as_float(kSentinelBits), __uint_as_float(kSentinelBits));
        const uint4 empty4 = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
        float4* v4 = reinterpret_cast<float4*>(vals);
        uint4* k4 = reinterpret_cast<uint4*>(keys);
#pragma unroll
        for (int g = 0; g < HK / 256; g++) { v4[g * 64 + lane] = sent4; k4[g * 64 + lane] = empty4; }
        if (AND) {
            uint32_t* m32 = reinterpret_cast<uint32_t*>(mcnt);
#pragma unroll
            for (int g = 0; g < HK / 256; g++) m32[g * 64 + lane] = 0;
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
        if (!it.whole) {
            const uint2* lst = seg.postings + tm.list_off;
            cur = list_lower_bound(lst, tm.count, it.doc_lo);
            end = list_lower_bound(lst, tm.count, it.doc_hi);
            if (end < cur) end = cur;
        }
    }
    // ---- the driver: the term with the most postings in this item's range (fixed for the item) ----
    uint32_t dl;
    {
        const uint32_t remv = end - cur;
        const uint32_t mx = wave_max_dpp(remv);
        dl = (uint32_t)__builtin_ctzll(__ballot(remv == mx && (uint32_t)lane < T) | (1ull << 63));
        if (dl >= T) dl = 0;
    }
    const uint32_t d_base = rdlane(base, dl);
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

    uint32_t lo = it.doc_lo;
    const uint32_t last_doc = it.doc_hi - 1;   // host guarantees doc_hi > doc_lo and doc_hi <= n_docs
    float theta = -__builtin_inff();
    uint32_t ncand = 0;
    uint32_t found_lane = 0;
    bool ge_mode = false;   // a shrink happened inside the current super-batch: ties with theta may still win on docId

    // offer (score, doc) of the lanes where `cond` holds to the candidate buffer
#define NS_OFFER(cond, scorev, docv)                                                               \
    {                                                                                              \
        bool qf_ = (cond) && (ge_mode ? ((scorev) >= theta) : ((scorev) > theta));                 \
        unsigned long long mask_ = __ballot(qf_);                                                  \
        if (mask_ != 0ull) {                                                                       \
            uint32_t n_ = (uint32_t)__popcll(mask_);                                               \
            if (ncand + n_ > (uint32_t)CB) {                                                       \
                ncand = wave_shrink(cand, ncand, theta, K, lane);                                  \
                ge_mode = true;                                                                    \
                qf_ = (cond) && ((scorev) >= theta);                                               \
                mask_ = __ballot(qf_);                                                             \
                n_ = (uint32_t)__popcll(mask_);                                                    \
            }                                                                                      \
            if (qf_) cand[ncand + lanes_below(mask_)] = make_key((scorev), (docv));                \
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
        const uint32_t nact_ = (uint32_t)__popcll(__ballot(rem_ > 0));                             \
        const float scale_ = (float)(FB - (int)nact_) * __builtin_amdgcn_rcpf((float)max(Rf, 1u)); \
        uint32_t w_ = 1u + (uint32_t)((float)rem_ * scale_);                                       \
        w_ = (w_ < rem_) ? w_ : rem_;                                                              \
        const bool probe_ = w_ < rem_;                                                             \
        const uint32_t pi_ = probe_ ? (base + cur + w_ - 1u) : 0u;                                 \
        const nat_u2 pv_ = postings[pi_];   /* unconditional load of a valid index: no branch */   \
        w_n = (rem_ > 0) ? w_ : 0u;                                                                \
        e_n = probe_ ? pv_.x : 0xFFFFFFFFu;                                                        \
    }
    NS_PLAN_FOREIGN();
    wave_sync();
    for (;;) {
        ge_mode = false;
        // ================= 1. foreign windows of this super-batch (planned one super-batch ahead) ==========
        const uint32_t w = w_n;
        const uint32_t e = e_n;
        const uint32_t incl = wave_incl_scan_dpp(w);
        const uint32_t total = rdlane(incl, 63);
        uint32_t hi = wave_min_dpp(e);   // every foreign posting with docId <= hi is inside its window
        hi = min(hi, last_doc);

        // ================= 2. foreign postings -> table =================
        uint32_t ftj[FE], fslot[FE];
        float fx[FE];
        bool fok[FE], fown[FE];
        uint32_t fdoc[FE];
        uint32_t tb_min = 0, tb_max = 0;
#pragma unroll
        for (int j = 0; j < FE; j++) { ftj[j] = 0; fslot[j] = 0; fx[j] = 0.0f; fok[j] = false; fown[j] = false; fdoc[j] = 0; }
        if (total > 0) {
            if ((uint32_t)lane < T) tab[lane] = make_uint4(idf_bits, wq_bits, base + cur - (incl - w), base + cur);
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
                pst[j] = postings[pidx[j]];
                nrm[j] = pnorm[pidx[j]];
                pst[j].x = inb ? pst[j].x : 0xFFFFFFFFu;   // docId ~0 is never <= hi
            }
            // ---- cursors: the first NOT-taken posting of a window publishes the new cursor ----
            uint32_t batch_consumed;
            {
                if ((uint32_t)lane < T) aux[lane] = base + cur + w;   // default: whole window consumed
                wave_sync();
                unsigned long long prev_last = 1ull;
#pragma unroll
                for (int j = 0; j < FE; j++) {
                    const bool take = pst[j].x <= hi;
                    fok[j] = take && (pst[j].x >= lo);   // docId < lo only for corrupt (unsorted) lists: consumed, not scored
                    fdoc[j] = pst[j].x;
                    const unsigned long long m = __ballot(take);
                    const bool prev_take = (((m << 1) | prev_last) >> lane) & 1ull;
                    prev_last = m >> 63;
                    const bool first_untaken = ((uint32_t)(j * 64 + lane) < total) && !take &&
                                               (prev_take || pidx[j] == tab[ftj[j]].w);
                    if (first_untaken) aux[ftj[j]] = pidx[j];
                }
                wave_sync();
                uint32_t c = 0;
                if ((uint32_t)lane < T && (uint32_t)lane != dl) {
                    const uint32_t ncur = aux[lane] - base;
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
            if (batch_consumed == 0) {   // only with corrupt lists (docIds beyond the range): skip the windows
                if ((uint32_t)lane != dl) { cur += w; if (cur > end) cur = end; }
                batch_consumed = total;
            }
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
#pragma unroll
            for (int j = 0; j < FE; j++) {
                const uint4 ent = tab[ftj[j]];
                const float tf = (float)pst[j].y;
                const float denom = tf + nrm[j];
                const float sc = (__uint_as_float(ent.x) * (tf * (1.2f + 1.0f))) / denom;
                fx[j] = __uint_as_float(ent.y) * sc;
            }
            // ---- claim one slot per distinct docId WITHOUT LDS atomics (integer and float LDS atomics
            //      are serialised per lane on gfx950): read the key; if the slot is free store our docId
            //      and read it back — a wave's LDS operations execute in order, so exactly one of the
            //      colliding docIds survives and everybody else moves on; equal docIds agree. ----
#pragma unroll
            for (int j = 0; j < FE; j++) {
                if ((uint32_t)(j * 64) >= total) continue;   // uniform
                uint32_t sl = (fdoc[j] * 2654435761u) >> (32 - LOG2HK);
                bool pending = fok[j];
                for (int round = 0; round < HK; round++) {
                    if (__ballot(pending) == 0ull) break;
                    uint32_t k = EMPTY;
                    if (pending) k = keys[sl];
                    if (pending && k == EMPTY) keys[sl] = fdoc[j];
                    wave_sync();
                    if (pending && k == EMPTY) k = keys[sl];
                    if (pending && k == fdoc[j]) pending = false;
                    if (pending) sl = (sl + 1) & (HK - 1);
                }
                fslot[j] = sl;
            }
            // ---- elect ONE owner per slot (several terms may hold the same doc): last store wins ----
#pragma unroll
            for (int j = 0; j < FE; j++)
                if (fok[j]) vals[fslot[j]] = __uint_as_float((uint32_t)(j * 64 + lane));
            wave_sync();
#pragma unroll
            for (int j = 0; j < FE; j++)
                if (fok[j]) fown[j] = __float_as_uint(vals[fslot[j]]) == (uint32_t)(j * 64 + lane);
            wave_sync();
#pragma unroll
            for (int j = 0; j < FE; j++)
                if (fown[j]) vals[fslot[j]] = __uint_as_float(kSentinelBits);
            wave_sync();
        }

        // term-ordered accumulation of foreign terms in [ta, tb] (read-add-write per term; docIds are
        // unique inside a term, so it is race-free; -0.0f is the exact additive identity)
#define NS_FOREIGN_RMW(ta, tb)                                                                     \
        for (uint32_t tt = (ta); tt <= (tb); tt++) {                                               \
            if (tt == dl) continue;                                                                \
            float old_[FE];                                                                        \
            _Pragma("unroll") for (int j = 0; j < FE; j++) {                                       \
                old_[j] = 0.0f;                                                                    \
                if (fok[j] && ftj[j] == tt) old_[j] = vals[fslot[j]];                              \
            }                                                                                      \
            _Pragma("unroll") for (int j = 0; j < FE; j++) {                                       \
                if (fok[j] && ftj[j] == tt) {                                                      \
                    vals[fslot[j]] = old_[j] + fx[j];                                              \
                    if (AND) mcnt[fslot[j]] = (uint8_t)(mcnt[fslot[j]] + 1);                       \
                }                                                                                  \
            }                                                                                      \
            wave_sync();                                                                           \
        }
        if (total > 0 && tb_min < dl) { NS_FOREIGN_RMW(tb_min, min(tb_max, dl - 1)); }

        // ================= 3. stream the driver's postings with docId <= hi =================
        bool driver_progress = false;
        for (;;) {
            const uint32_t remd = d_end - d_cur;
            if (remd == 0) break;
            const uint32_t n = min(remd, (uint32_t)(DE * 64));
            nat_u2 ps[DE];
            float nr[DE];
#pragma unroll
            for (int j = 0; j < DE; j++) {
                const uint32_t p = (uint32_t)(j * 64 + lane);
                const uint32_t idx = d_base + d_cur + ((p < n) ? p : 0u);
                ps[j] = postings[idx];
                nr[j] = pnorm[idx];
                ps[j].x = (p < n) ? ps[j].x : 0xFFFFFFFFu;
            }
            uint32_t cnt = 0;
            float dx[DE];
            bool dok[DE];
#pragma unroll
            for (int j = 0; j < DE; j++) {
                const bool take = ps[j].x <= hi;
                cnt += (uint32_t)__popcll(__ballot(take));
                dok[j] = take && (ps[j].x >= lo);
                const float tf = (float)ps[j].y;
                const float denom = tf + nr[j];
                const float sc = (d_idf * (tf * (1.2f + 1.0f))) / denom;
                dx[j] = d_wq * sc;
            }
            // does any foreign doc of this super-batch fall into the doc range of this round?  (usually
            // not when the driver is much denser than the foreign lists: then the lookups are skipped)
            bool any_foreign_here = total > 0;
            if (FB <= 64 && total > 0) {   // only worth testing for the thin-foreign class
                const uint32_t rfirst = rdlane(ps[0].x, 0);
                uint32_t rlast = 0;
#pragma unroll
                for (int j = 0; j < DE; j++)
                    if (((n - 1u) >> 6) == (uint32_t)j) rlast = rdlane(ps[j].x, (n - 1u) & 63u);   // uniform
                bool here = false;
#pragma unroll
                for (int j = 0; j < FE; j++) here = here || (fok[j] && fdoc[j] >= rfirst && fdoc[j] <= rlast);
                any_foreign_here = __ballot(here) != 0ull;
            }
            if (any_foreign_here) {
                // one lookup per posting; a hit joins the table accumulation at the driver's place in
                // the term order (foreign terms before it are already in, those after it follow)
#pragma unroll
                for (int j = 0; j < DE; j++) {
                    if ((uint32_t)(j * 64) >= n) continue;   // uniform
                    uint32_t sl = (ps[j].x * 2654435761u) >> (32 - LOG2HK);
                    bool pending = dok[j];
                    bool hit = false;
                    for (int round = 0; round < HK; round++) {
                        if (__ballot(pending) == 0ull) break;
                        uint32_t k = EMPTY;
                        if (pending) k = keys[sl];
                        if (pending && k == ps[j].x) { hit = true; pending = false; }
                        if (pending && k == EMPTY) pending = false;
                        if (pending) sl = (sl + 1) & (HK - 1);
                    }
                    if (hit) {
                        vals[sl] = vals[sl] + dx[j];
                        if (AND) mcnt[sl] = (uint8_t)(mcnt[sl] + 1);
                        dok[j] = false;   // scored through the table's owner
                    }
                }
            }
            // private postings: no other term has the doc: score == 0.0f + w*s == w*s exactly
#pragma unroll
            for (int j = 0; j < DE; j++) {
                const bool scored = dok[j] && (!AND || T == 1);   // conjunctive extension: one term alone never qualifies
                found_lane += scored ? 1u : 0u;
                NS_OFFER(scored, dx[j], ps[j].x);
            }
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
                fin[j] = 0.0f;
                scored[j] = fown[j];
                if (fown[j]) {
                    fin[j] = vals[fslot[j]];
                    if (AND) scored[j] = (mcnt[fslot[j]] == (uint8_t)T);   // conjunctive extension: every term ref hit the doc
                }
            }
            wave_sync();
#pragma unroll
            for (int j = 0; j < FE; j++) {
                if (fown[j]) {   // the owner resets the slot for the next super-batch
                    vals[fslot[j]] = __uint_as_float(kSentinelBits);
                    keys[fslot[j]] = EMPTY;
                    if (AND) mcnt[fslot[j]] = 0;
                }
                found_lane += scored[j] ? 1u : 0u;
                NS_OFFER(scored[j], fin[j], fdoc[j]);
            }
            wave_sync();
        }
        if (ncand > (uint32_t)(CB / 2)) ncand = wave_shrink(cand, ncand, theta, K, lane);

        if (hi >= last_doc) break;
        if (Rf == 0 && d_cur >= d_end) break;
        if (total == 0 && !driver_progress) break;   // only with corrupt lists: nothing can advance
        lo = hi + 1;
    }
#undef NS_FOREIGN_RMW
#undef NS_OFFER
#undef NS_PLAN_FOREIGN

    // ---- this item's top-K ----
    wave_sync();
    ncand = wave_shrink(cand, ncand, theta, K, lane);
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
    uint32_t found = found_lane;
    found += dpp_mov<0x111, 0xf>(0u, found);
    found += dpp_mov<0x112, 0xf>(0u, found);
    found += dpp_mov<0x114, 0xf>(0u, found);
    found += dpp_mov<0x118, 0xf>(0u, found);
    found += dpp_mov<0x142, 0xa>(0u, found);
    found += dpp_mov<0x143, 0xc>(0u, found);
    if (lane == 63) {
        out_nhits[it.out_slot] = n;
        out_found[it.out_slot] = (uint64_t)found;
    }
}

template <int HK, int FB, bool AND>
__global__ void __launch_bounds__(256) k_dscore(const DevWItem* __restrict__ items, uint32_t n_items,
                                                const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                                Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                                uint64_t* __restrict__ out_found, uint32_t K) {
    constexpr int WPB = 4;                 // independent waves per workgroup
    __shared__ __attribute__((aligned(16))) float s_vals[WPB][HK];
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[WPB][HK];
    __shared__ __attribute__((aligned(16))) uint8_t s_mcnt[WPB][AND ? HK : 16];   // AND: term refs that hit the slot
    __shared__ uint64_t s_cand[WPB][256];
    __shared__ __attribute__((aligned(16))) uint4 s_tab[WPB][64];   // per term: {idf, qweight, first posting - excl prefix, first posting}
    __shared__ uint32_t s_aux[WPB][64];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const uint32_t item_idx = blockIdx.x * WPB + wave;
    if (item_idx >= n_items) return;   // whole wave leaves; there is no workgroup barrier in this kernel
    const DevWItem it = items[item_idx];
    dscore_body<HK, FB, AND>(it, terms, segs, s_vals[wave], s_keys[wave], s_mcnt[wave], s_cand[wave], s_tab[wave], s_aux[wave],
                             out_hits, out_nhits, out_found, K, lane);
}

}  // namespace ns

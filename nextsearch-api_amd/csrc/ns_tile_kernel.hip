// k_tscore — wave-private "doc-tile" scoring kernel for DENSE term groups (several frequent lists).
//
// One WAVE scores one work item = one (query, segment) term group (<= 64 terms) over a doc range,
// tile by tile of TD docs.  The accumulator is a direct-mapped fp32 table in LDS (slot = docId -
// tile start): no hashing, no keys, no lookups, no ownership.  Per tile, the terms are applied one
// after the other in query-term order (the fp32 accumulation order of src/api_engine.cpp:449,480):
// the term's postings with docId inside the tile stream through in rounds of <= 256 (the round
// size follows the expected number of postings in the tile), each posting does a plain
// read-add-write on its slot (docIds are unique inside a term, so it is race-free; LDS float atomics
// are ~3 clk per lane on gfx950 and are not used).  Then the tile is read back from LDS: touched
// slots are counted (`found`, :495), offered as candidates when above theta (:485-492), and reset.
//
// Cost per tile ~ 100 + sum over terms (70 + 0.6 x postings): the host routes a group here when
// that beats the driver-stream kernel (k_dscore), i.e. when the non-driver lists are dense.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ns_internal.h"
#include "ns_wave_kernel.hip"
#include "ns_driver_kernel.hip"
#include "ns_prune_kernel.hip"
#include "ns_merge_kernel.hip"

namespace ns {

#ifdef NS_COUNT
__device__ unsigned long long g_ns_tcnt[12];   // diagnostic build: event counts of the doc-tile body (tools/dbg/count_run.py)
#endif

template <int TD, bool AND, int CB = 256, bool IMP = false>
__device__ __forceinline__ void tscore_body(const DevWItem& it, const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                            float* vals, uint8_t* mcnt, uint64_t* cand,
                                            Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                            uint64_t* __restrict__ out_found, uint32_t K, const int lane) {
    // CB: candidate buffer entries, a power of two >= K + 64
    constexpr int E = 4;                   // postings per lane per round
    constexpr int NG = TD / 256;           // float4 groups per lane in the tile read-back
    static_assert(TD == 512 || TD == 1024 || TD == 2048, "TD must be 512, 1024 or 2048");

    const DevSeg seg = segs[it.seg];
    const uint32_t T = it.term_count;
    const bool fast_div = (__builtin_amdgcn_readfirstlane((int)it.whole) & 8) != 0;   // see ns_div_short
    const gp_u2 postings = (gp_u2)seg.postings;
    const gp_f32 pnorm = (gp_f32)seg.pnorm;
    const gp_u2 stream = IMP ? (gp_u2)seg.impacts : postings;   // IMP: {docId, precomputed term score bits} (see dscore_body)
    // Skip grid (DevWItem::whole bit 5; the host sets it only for TD == kSkipDocs and a doc_lo on the grid): tiles are the
    // cells of the segment's skip tables, and a term that has a table takes EXACTLY its postings of the tile — entry[tile]
    // .. entry[tile + 1] — instead of estimating a round and finding the tile's end by comparing docIds: no partly used
    // rounds, no cursor search.  Terms without a table (short lists) keep their cursors, in the same tiles.
    const bool grid = TD == (int)kSkipDocs && (__builtin_amdgcn_readfirstlane((int)it.whole) & 32) != 0;
    const gp_u32 skips = (gp_u32)seg.skips;

    const float4 sent4 = make_float4(__uint_as_float(kTileEmptyBits), __uint_as_float(kTileEmptyBits),
                                     __uint_as_float(kTileEmptyBits), __uint_as_float(kTileEmptyBits));   // see ns_internal.h
    float4* v4 = reinterpret_cast<float4*>(vals);
#pragma unroll
    for (int g = 0; g < NG; g++) v4[g * 64 + lane] = sent4;
    if (AND) {
        uint32_t* m32 = reinterpret_cast<uint32_t*>(mcnt);
#pragma unroll
        for (int g = 0; g < NG; g++) m32[g * 64 + lane] = 0;
    }

    // ---- lane t owns term t: absolute posting cursor, end, and the docId at the cursor (~0: exhausted).
    //      Posting indices are 32-bit: upload rejects segments of >= 2^32 postings. ----
    //      A term on the skip grid: `cur` .. `end` are its postings of the CURRENT tile, `nxt` the end of the next tile's
    //      (read one tile ahead), `skb` the index of its table entry for tile 0; its `nd` stays ~0. ----
    uint32_t cur = 0, end = 0, nd = 0xFFFFFFFFu, idf_bits = 0, wq_bits = 0, nxt = 0, skb = 0;
    bool on_grid = false;
    if ((uint32_t)lane < T) {
        const DevTerm tm = terms[it.term_begin + lane];
        idf_bits = __float_as_uint(tm.idf);
        wq_bits = __float_as_uint(tm.weight);
        if (grid && tm.skip != 0u) {
            on_grid = true;
            skb = tm.skip - 1u;
            const uint32_t t0 = skb + it.doc_lo / (uint32_t)TD;
            cur = skips[t0];
            end = skips[t0 + 1u];
            nxt = skips[t0 + 2u];
            if (end < cur) end = cur;   // tables are checked when they are built; a wrapped count must never happen
        } else {
            end = tm.count;
            if (!(it.whole & 1u)) {
                const uint2* lst = seg.postings + tm.list_off;
                list_range(lst, tm.count, it.doc_lo, it.doc_hi, seg.n_docs, cur, end);
                if (end < cur) end = cur;
            }
            cur += (uint32_t)tm.list_off;
            end += (uint32_t)tm.list_off;
            if (cur < end) { const nat_u2 pv = postings[cur]; nd = pv.x; }
        }
    }
    const uint64_t gridm = grid ? wballot(on_grid) : 0ull;

    const uint32_t last_doc = it.doc_hi - 1;   // host guarantees doc_hi > doc_lo and doc_hi <= n_docs
    float theta = -__builtin_inff();
    uint32_t ncand = 0;
    uint32_t nsorted = 0;   // leading candidates already in descending order (left by the last shrink)
    uint32_t found_s = 0;   // wave-uniform count (popcounts of ballots)
#ifdef NS_COUNT
    unsigned long long tc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define NS_TCNT(i, v) tc_[(i)] += (unsigned long long)(v)
    NS_TCNT(0, 1);
    const unsigned long long tc_t0_ = __builtin_readcyclecounter();
#else
#define NS_TCNT(i, v)
#endif
    wave_sync();

    // One round = up to E*64 postings of one term, loaded with a scalar base + a fixed lane offset (lanes
    // past n read the next list or the buffer's padding and are masked).  One round is in flight per wave: loading the
    // round expected next into a second register set, or into the same registers once they are dead, was measured slower
    // (fewer waves; the compiler waits for loop-carried loads at the back-edge — DESIGN §5, ab4 / ab11).
    nat_u2 ps[E];
    float nr[E];
#define NS_ISSUE(PS, NR, start, nn_)                                                               \
    {                                                                                              \
        const gp_u2 sp_ = stream + (start);                                                        \
        const gp_f32 np_ = pnorm + (start);                                                        \
        _Pragma("unroll") for (int j = 0; j < E; j++) {                                            \
            if ((uint32_t)(j * 64) >= (nn_)) continue;   /* uniform: chunk beyond this round */    \
            PS[j] = sp_[j * 64 + lane];                                                            \
            if (!IMP) NR[j] = np_[j * 64 + lane];                                                  \
        }                                                                                          \
    }
#define NS_ROUND_SIZE(rem_, want_) min(min((rem_), (uint32_t)(E * 64)), max((want_), 64u))
#pragma unroll
    for (int j = 0; j < E; j++) { ps[j] = nat_u2{0xFFFFFFFFu, 0u}; nr[j] = 1.0f; }

    // ---- tile header: the next tile starts at the first doc that still has a posting (empty doc space costs nothing) ----
    uint32_t lo = it.doc_lo, hi = 0;
    float frac = 0.0f;
    uint64_t act = 0ull;
    bool done = false;
#define NS_TILE_HEADER()                                                                           \
    if (gridm != 0ull) {   /* grid cells one after the other (lo stays on the grid) */             \
        done = lo > last_doc;                                                                      \
        if (!done) {                                                                               \
            hi = (last_doc - lo >= (uint32_t)TD) ? (lo + (uint32_t)(TD - 1)) : last_doc;           \
            frac = (float)(hi - lo + 1u) * __builtin_amdgcn_rcpf((float)(last_doc - lo + 1u));     \
            act = wballot(nd <= hi) | (wballot(end > cur) & gridm);                                \
        }                                                                                          \
    } else {                                                                                       \
        const uint32_t mind_ = wave_min_dpp(nd);                                                   \
        done = mind_ > last_doc;                                                                   \
        if (!done) {                                                                               \
            if (mind_ > lo) lo = mind_;                                                            \
            hi = (last_doc - lo >= (uint32_t)TD) ? (lo + (uint32_t)(TD - 1)) : last_doc;           \
            /* fraction of the remaining doc range covered by this tile: sizes the rounds */       \
            frac = (float)(hi - lo + 1u) * __builtin_amdgcn_rcpf((float)(last_doc - lo + 1u));     \
            act = wballot(nd <= hi);                                                               \
        }                                                                                          \
    }
    NS_TILE_HEADER();
    while (!done) {
        const uint32_t tile_lo = lo, tile_hi = hi;
        const bool touched = act != 0ull;            // a grid cell may hold no posting of the group at all
        NS_TCNT(1, 1);                               // tiles
        NS_TCNT(2, __popcll(act));                   // (term, tile) visits
        // ---- the terms that have postings in this tile, in query order (the fp32 accumulation order) ----
        while (act != 0ull) {
            const uint32_t t = (uint32_t)__builtin_ctzll(act);
            act &= act - 1ull;
            uint32_t s_cur = rdlane(cur, t);
            const uint32_t s_end = rdlane(end, t);
            const float idf = __uint_as_float(rdlane(idf_bits, t));
            const float wq = __uint_as_float(rdlane(wq_bits, t));
            const bool exact = ((gridm >> t) & 1ull) != 0ull;   // [s_cur, s_end) are the term's postings of this tile, all of them
            uint32_t want = exact ? (uint32_t)(E * 64) : 16u + (uint32_t)((float)(s_end - s_cur) * frac * 1.125f);
            uint32_t s_nd = 0xFFFFFFFFu;
            for (;;) {
                const uint32_t remd = s_end - s_cur;
                if (remd == 0) break;
                if (exact && remd >= (uint32_t)(E * 64)) {
                    // A FULL round on the skip grid: 256 postings that are all in this tile.  No lane masks, no exec
                    // games, no per-chunk branches: the arithmetic of the general round below with every lane live.
                    // (Measured next to it: ALL rounds of such a term as straight-line code, lanes past the segment's end
                    // parked on dummy slots — fewer instructions, but slower on mixed batches; profiles/r02/ab.)
#ifdef NS_COUNT
                    const unsigned long long tr0_ = __builtin_readcyclecounter();
#endif
                    NS_ISSUE(ps, nr, s_cur, (uint32_t)(E * 64));
#ifdef NS_COUNT
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const unsigned long long tr1_ = __builtin_readcyclecounter();
                    NS_TCNT(8, tr1_ - tr0_); NS_TCNT(9, 1);
#endif
                    NS_TCNT(3, 1); NS_TCNT(4, E); NS_TCNT(5, E * 64);
                    float xq[E], oldq[E];
                    uint32_t slq[E];
#pragma unroll
                    for (int j0 = 0; j0 < E; j0 += 2) {
                        float num_[2], den_[2], q_[2];
#pragma unroll
                        for (int jj = 0; jj < 2; jj++) {
                            const float tf = (float)ps[j0 + jj].y;
                            den_[jj] = tf + nr[j0 + jj];
                            num_[jj] = idf * (tf * (1.2f + 1.0f));
                        }
                        if (IMP) { q_[0] = __uint_as_float(ps[j0].y); q_[1] = __uint_as_float(ps[j0 + 1].y); }
                        else ns_div_n<2>(q_, num_, den_, fast_div);
#pragma unroll
                        for (int jj = 0; jj < 2; jj++) {
                            xq[j0 + jj] = q_[jj];
                            slq[j0 + jj] = (ps[j0 + jj].x - tile_lo) & (uint32_t)(TD - 1);
                            oldq[j0 + jj] = vals[slq[j0 + jj]];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < E; j++) {
                        const bool fresh_ = __float_as_uint(oldq[j]) == kTileEmptyBits;
                        if (!AND) found_s += (uint32_t)__popcll(wballot(fresh_));
                        oldq[j] = fresh_ ? 0.0f : oldq[j];
                    }
#pragma unroll
                    for (int j = 0; j < E; j++) {
                        vals[slq[j]] = oldq[j] + wq * xq[j];
                        if (AND) mcnt[slq[j]] = (uint8_t)(mcnt[slq[j]] + 1);
                    }
                    s_cur += (uint32_t)(E * 64);
#ifdef NS_COUNT
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    NS_TCNT(11, __builtin_readcyclecounter() - tr0_);
#endif
                    continue;
                }
                const uint32_t n = NS_ROUND_SIZE(remd, want);
                NS_ISSUE(ps, nr, s_cur, n);
                NS_TCNT(3, 1);                       // rounds
                NS_TCNT(4, (n + 63) / 64);           // chunks loaded
                const bool expect_more = want > n;
                if (!exact) want = expect_more ? (want - n) : 64u;   // exact: every round is as large as what is left allows
                uint32_t cnt = 0;
                uint64_t takem[E];
                float x[E];
                float old[E];
                uint32_t slot[E];
#pragma unroll
                for (int j = 0; j < E; j++) { takem[j] = 0ull; x[j] = 0.0f; old[j] = 0.0f; slot[j] = 0; }
                // A round holds 1..E chunks of 64 postings; the work is done per PAIR of chunks (the compiler
                // packs the two BM25 evaluations into v_pk_* instructions), and the second pair only if it has postings.
#define NS_TILE_PAIR(J0, NJ)                                                                         \
                {   /* src/api_engine.cpp:477-480, operation for operation (k1 + 1.0f == 0x400CCCCD) */ \
                    float num_[NJ], den_[NJ], q_[NJ];                                              \
                    _Pragma("unroll") for (int jj = 0; jj < (NJ); jj++) {                          \
                        const int j = (J0) + jj;                                                   \
                        const uint32_t left = (n > (uint32_t)(j * 64)) ? (n - (uint32_t)(j * 64)) : 0u;   /* scalar */ \
                        const uint64_t nmask = (left >= 64u) ? ~0ull : ((1ull << left) - 1ull);    \
                        takem[j] = exact ? nmask : (wballot(ps[j].x <= tile_hi) & nmask);          \
                        cnt += (uint32_t)__popcll(takem[j]);                                       \
                        const float tf = (float)ps[j].y;                                           \
                        den_[jj] = tf + nr[j];                                                     \
                        num_[jj] = idf * (tf * (1.2f + 1.0f));                                     \
                    }                                                                              \
                    if (IMP) { _Pragma("unroll") for (int jj = 0; jj < (NJ); jj++) q_[jj] = __uint_as_float(ps[(J0) + jj].y); } \
                    else ns_div_n<NJ>(q_, num_, den_, fast_div);                                   \
                    _Pragma("unroll") for (int jj = 0; jj < (NJ); jj++) x[(J0) + jj] = q_[jj];       \
                    /* read-add-write on the doc's slot; all reads of the round first (docIds of one term are */ \
                    /* distinct).  The slot index is masked: a corrupt (unsorted) list cannot leave the tile. */ \
                    _Pragma("unroll") for (int jj = 0; jj < (NJ); jj++) {                          \
                        const int j = (J0) + jj;                                                   \
                        slot[j] = (ps[j].x - tile_lo) & (uint32_t)(TD - 1);                        \
                        if (takem[j] == 0ull) continue;   /* uniform */                            \
                        old[j] = vals[slot[j]];   /* every lane reads (the slot index is masked, the value of a lane that does not take is dropped) */ \
                        /* `found` (:495) counts a doc when its slot is touched for the first time (OR mode; the */ \
                        /* conjunctive extension counts in the read-back, where the per-doc term counts are known); */ \
                        /* a first touch starts from the reference's +0.0f (:480), whatever the contribution is */ \
                        const bool fresh_ = __float_as_uint(old[j]) == kTileEmptyBits;             \
                        if (!AND) found_s += (uint32_t)__popcll(takem[j] & wballot(fresh_));       \
                        old[j] = fresh_ ? 0.0f : old[j];                                           \
                    }                                                                              \
                }
                if (n <= 64u) NS_TILE_PAIR(0, 1) else NS_TILE_PAIR(0, 2);
                if (n > 192u) NS_TILE_PAIR(2, 2) else if (n > 128u) NS_TILE_PAIR(2, 1);
#undef NS_TILE_PAIR
#pragma unroll
                for (int j = 0; j < E; j++) {
                    if (takem[j] == 0ull) continue;   // uniform
                    if (__builtin_amdgcn_inverse_ballot_w64(takem[j])) {
                        vals[slot[j]] = old[j] + wq * x[j];
                        if (AND) mcnt[slot[j]] = (uint8_t)(mcnt[slot[j]] + 1);
                    }
                }
                NS_TCNT(5, cnt);                     // postings taken
                s_cur += cnt;
                if (cnt < n) {   // reached the end of the tile: the first posting not taken is the term's next doc
                    // select the chunk with two uniform v_cndmask levels, then ONE readlane
                    const uint32_t jc = cnt >> 6;
                    const uint32_t x01 = (jc & 1u) ? ps[1].x : ps[0].x;
                    const uint32_t x23 = (jc & 1u) ? ps[3].x : ps[2].x;
                    s_nd = rdlane((jc & 2u) ? x23 : x01, cnt & 63u);
                    break;
                }
            }
            if ((uint32_t)lane == t) { cur = s_cur; nd = s_nd; }
            wave_sync();   // the next term's read-add-writes follow this term's
        }

        // ---- header of the NEXT tile (its skip entries were read one tile ago) ----
        if (tile_hi >= last_doc) { done = true; }
        else {
            lo = tile_hi + 1u;
            if (gridm != 0ull) {   // the next cell: its postings end where the entry read one tile ago says; read the one after
                if (on_grid) {
                    cur = end;
                    end = nxt < cur ? cur : nxt;
                    nxt = skips[skb + lo / (uint32_t)TD + 2u];
                }
            }
            NS_TILE_HEADER();
        }

        // ---- read the tile back: candidates, reset (and, for the conjunctive extension, found) ----
        bool ge_mode = false;   // after a shrink INSIDE this tile, ties with theta may still win on docId
#ifdef NS_COUNT
        const unsigned long long trb0_ = __builtin_readcyclecounter();
#endif
        if (touched) {
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const float4 q = v4[g * 64 + lane];
            const float vv[4] = {q.x, q.y, q.z, q.w};
            v4[g * 64 + lane] = sent4;
            uint32_t cw = 0;
            if (AND) { cw = reinterpret_cast<const uint32_t*>(mcnt)[g * 64 + lane]; reinterpret_cast<uint32_t*>(mcnt)[g * 64 + lane] = 0; }
            // An untouched slot holds a NaN, which never compares above (or equal to) theta, and fmax ignores it:
            // one max + one compare per four slots decides whether anything here can be offered.
            if (!AND) {
                const float mx = __builtin_fmaxf(__builtin_fmaxf(vv[0], vv[1]), __builtin_fmaxf(vv[2], vv[3]));
                if ((ge_mode ? wballot(mx >= theta) : wballot(mx > theta)) == 0ull) continue;
            }
            uint64_t scm[4];
            uint64_t anyq = 0ull;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                scm[c] = wballot(__float_as_uint(vv[c]) != kTileEmptyBits);
                if (AND) {
                    scm[c] &= wballot(((cw >> (8 * c)) & 0xFFu) == T);   // conjunctive extension
                    found_s += (uint32_t)__popcll(scm[c]);
                }
                anyq |= scm[c] & wballot(vv[c] > theta);
            }
            if (anyq != 0ull) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    uint64_t mask = scm[c] & (ge_mode ? wballot(vv[c] >= theta) : wballot(vv[c] > theta));
                    if (mask != 0ull) {
                        uint32_t n = (uint32_t)__popcll(mask);
                        if (ncand + n > (uint32_t)CB) {
                            ncand = wave_shrink_cb<CB>(cand, ncand, nsorted, theta, K, lane);
                            ge_mode = true;
                            mask = scm[c] & wballot(vv[c] >= theta);
                            n = (uint32_t)__popcll(mask);
                        }
                        if (__builtin_amdgcn_inverse_ballot_w64(mask))
                            cand[ncand + lanes_below(mask)] = make_key(vv[c], tile_lo + (uint32_t)((g * 64 + lane) * 4 + c));
                        ncand += n;
                    }
                }
            }
        }
        }
        wave_sync();
        if (ncand > (uint32_t)(CB - 64)) ncand = wave_shrink_cb<CB>(cand, ncand, nsorted, theta, K, lane);   // keep room for one more step of offers
#ifdef NS_COUNT
        NS_TCNT(10, __builtin_readcyclecounter() - trb0_);
#endif
    }
#undef NS_ISSUE
#undef NS_ROUND_SIZE
#undef NS_TILE_HEADER

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
    NS_TCNT(6, T);
    NS_TCNT(7, __builtin_readcyclecounter() - tc_t0_);
    if (lane == 0)
        for (int i = 0; i < 12; i++) if (tc_[i]) atomicAdd(&g_ns_tcnt[i], tc_[i]);
#endif
#undef NS_TCNT
}

template <int TD, bool AND>
__global__ void __launch_bounds__(256) k_tscore(const DevWItem* __restrict__ items, uint32_t n_items,
                                                const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                                Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                                uint64_t* __restrict__ out_found, uint32_t K) {
    constexpr int WPB = 4;                 // independent waves per workgroup
    __shared__ __attribute__((aligned(16))) float s_vals[WPB][TD];
    __shared__ __attribute__((aligned(16))) uint8_t s_mcnt[WPB][AND ? TD : 16];   // AND: term refs that hit the slot
    __shared__ uint64_t s_cand[WPB][256];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: the item is fetched with scalar loads
    const int lane = threadIdx.x & 63;
    const uint32_t item_idx = blockIdx.x * WPB + wave;
    if (item_idx >= n_items) return;   // whole wave leaves; there is no workgroup barrier in this kernel
    const DevWItem it = items[item_idx];
    tscore_body<TD, AND>(it, terms, segs, s_vals[wave], s_mcnt[wave], s_cand[wave], out_hits, out_nhits, out_found, K, lane);
}

// Unified scoring kernel (the default): every wave picks, per work item, the body that suits the
// item's mix of posting lists — driver stream + foreign table (dscore_body) or doc tiles for very
// dense groups (tscore_body; DevWItem::whole bit 1).  One launch, one LDS budget: the tile body's
// 2*HK-slot table aliases the driver body's HK values + HK keys.
// TMAX: most terms a group may have in this instantiation (16: smaller term tables, 3 KB less LDS per
// workgroup; 64: the general case, launched only when needed).  An 8-entry instantiation reaches 7
// workgroups per CU (23.2 KB LDS, 69 VGPRs) and measured 1.5% slower: the kernel is VALU-bound.
// Waves per workgroup of k_uscore.  The waves of a workgroup are independent (no barrier, private LDS
// slices), but a workgroup's wave slots and LDS are only released when its LAST wave ends: with 4 items of
// unequal length per workgroup ~12 % of the wave slots sat idle mid-kernel.  One item per workgroup.
constexpr int kUscoreWavesPerBlock = 1;

// PK != 0: the driver stream of THIN groups (one list dominates: long runs of whole blocks) reads the packed posting blocks
// (ns_segment_build_packed) instead of {docId, tf}; norms from the fp32 norm stream (PK == 1) or through the blocks' 16-bit
// norm index (PK == 2).  General groups (super-batches of ~1.4 blocks: most rounds would be partial blocks, measured 9 %
// slower), foreign windows and doc tiles keep reading the raw stream.
template <int HK, int FB, bool AND, int CB, int TMAX, bool IMP = false, int PK = 0>
__global__ void __launch_bounds__(64 * kUscoreWavesPerBlock) __attribute__((amdgpu_waves_per_eu(6, 8))) k_uscore(const DevWItem* __restrict__ items, uint32_t n_items,
                                                const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                                Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                                uint64_t* __restrict__ out_found, uint32_t K) {
    constexpr int WPB = kUscoreWavesPerBlock;
    __shared__ __attribute__((aligned(16))) uint32_t s_tbl[WPB][2 * HK];                 // HK/2 buckets of 4 entries, or one 2*HK-doc tile
    __shared__ __attribute__((aligned(16))) float s_vals[WPB][FB];                       // driver body: one accumulator per foreign posting
    __shared__ __attribute__((aligned(16))) uint8_t s_mcnt[WPB][AND ? 2 * HK : 16];
    __shared__ uint64_t s_cand[WPB][CB];
    __shared__ __attribute__((aligned(16))) uint4 s_tab[WPB][TMAX];
    __shared__ uint32_t s_aux[WPB][TMAX];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: the item is fetched with scalar loads
    const int lane = threadIdx.x & 63;
    const uint32_t item_idx = blockIdx.x * WPB + wave;
    if (item_idx >= n_items) return;   // whole wave leaves; there is no workgroup barrier in this kernel
    DevWItem it = items[item_idx];
    const bool tiles = (it.whole & 2u) != 0;
    const bool thin = (it.whole & 4u) != 0;   // the non-driver lists are thin: smallest foreign budget
    const bool pruned = (it.whole & 128u) != 0;   // single-term item with block maxima (ns_ctx_use_pruning)
    const bool merge2 = (it.whole & 256u) != 0;   // two-list group: the merge body (ns_ctx_use_merge)
    it.whole &= 121u;   // bit 0: whole segment, bit 3: short division, bit 4: signed inputs, bits 5 / 6: skip grid (doc tiles / range ends)
    if (pruned)
        pscore_body<AND, CB>(it, terms, segs, s_cand[wave], out_hits, out_nhits, out_found, K, lane);
    else if (merge2)
        mscore_body<AND, CB, IMP>(it, terms, segs, s_tbl[wave], reinterpret_cast<float*>(s_tbl[wave] + 256), s_cand[wave],
                             out_hits, out_nhits, out_found, K, lane);
    else if (thin)
        dscore_body<HK / 2, 64, AND, CB, IMP, PK>(it, terms, segs, s_tbl[wave], s_vals[wave], s_mcnt[wave], s_cand[wave],
                                 s_tab[wave], s_aux[wave], out_hits, out_nhits, out_found, K, lane);
    else if (tiles)
        tscore_body<2 * HK, AND, CB, IMP>(it, terms, segs, reinterpret_cast<float*>(s_tbl[wave]), s_mcnt[wave], s_cand[wave],
                                 out_hits, out_nhits, out_found, K, lane);
    else
        dscore_body<HK / 2, FB, AND, CB, IMP, 0>(it, terms, segs, s_tbl[wave], s_vals[wave], s_mcnt[wave], s_cand[wave],
                                 s_tab[wave], s_aux[wave], out_hits, out_nhits, out_found, K, lane);
}

}  // namespace ns

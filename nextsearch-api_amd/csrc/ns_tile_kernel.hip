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

namespace ns {

template <int TD, bool AND, int CB = 256>
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

    const float4 sent4 = make_float4(__uint_as_float(kSentinelBits), __uint_as_float(kSentinelBits),
                                     __uint_as_float(kSentinelBits), __uint_as_float(kSentinelBits));
    float4* v4 = reinterpret_cast<float4*>(vals);
#pragma unroll
    for (int g = 0; g < NG; g++) v4[g * 64 + lane] = sent4;
    if (AND) {
        uint32_t* m32 = reinterpret_cast<uint32_t*>(mcnt);
#pragma unroll
        for (int g = 0; g < NG; g++) m32[g * 64 + lane] = 0;
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
            const uint2* lst = seg.postings + tm.list_off;
            cur = list_lower_bound(lst, tm.count, it.doc_lo);
            end = list_lower_bound(lst, tm.count, it.doc_hi);
            if (end < cur) end = cur;
        }
    }

    const uint32_t last_doc = it.doc_hi - 1;   // host guarantees doc_hi > doc_lo and doc_hi <= n_docs
    float theta = -__builtin_inff();
    uint32_t ncand = 0;
    uint32_t found_lane = 0;
    wave_sync();

    for (uint32_t lo = it.doc_lo;; lo += (uint32_t)TD) {
        const uint32_t hi = min(lo + (uint32_t)(TD - 1), last_doc);
        // fraction of the remaining doc range covered by this tile: sizes the rounds
        const float frac = (float)(hi - lo + 1u) * __builtin_amdgcn_rcpf((float)(last_doc - lo + 1u));

        // ---- terms in query order ----
        for (uint32_t t = 0; t < T; t++) {
            const uint32_t s_base = rdlane(base, t);
            uint32_t s_cur = rdlane(cur, t);
            const uint32_t s_end = rdlane(end, t);
            if (s_cur >= s_end) continue;
            const float idf = __uint_as_float(rdlane(idf_bits, t));
            const float wq = __uint_as_float(rdlane(wq_bits, t));
            uint32_t want = 16u + (uint32_t)((float)(s_end - s_cur) * frac * 1.125f);
            for (;;) {
                const uint32_t remd = s_end - s_cur;
                if (remd == 0) break;
                const uint32_t n = min(min(remd, (uint32_t)(E * 64)), max(want, 64u));
                want = (want > n) ? (want - n) : 64u;
                nat_u2 ps[E];
                float nr[E];
#pragma unroll
                for (int j = 0; j < E; j++) {
                    ps[j] = nat_u2{0xFFFFFFFFu, 0u};
                    nr[j] = 1.0f;
                    if ((uint32_t)(j * 64) >= n) continue;   // uniform: chunk beyond this round
                    const uint32_t p = (uint32_t)(j * 64 + lane);
                    const uint32_t idx = s_base + s_cur + ((p < n) ? p : 0u);
                    ps[j] = postings[idx];
                    nr[j] = pnorm[idx];
                    ps[j].x = (p < n) ? ps[j].x : 0xFFFFFFFFu;   // docId ~0 is never <= hi
                }
                uint32_t cnt = 0;
                float x[E], old[E];
                bool ok[E];
#pragma unroll
                for (int j = 0; j < E; j++) {
                    ok[j] = false; x[j] = 0.0f; old[j] = 0.0f;
                    if ((uint32_t)(j * 64) >= n) continue;   // uniform
                    const bool take = ps[j].x <= hi;
                    cnt += (uint32_t)__popcll(__ballot(take));
                    // docId < lo only for corrupt (unsorted) lists: consumed, not scored
                    ok[j] = take && (ps[j].x >= lo);
                    // src/api_engine.cpp:477-480, operation for operation (k1 + 1.0f == 0x400CCCCD)
                    const float tf = (float)ps[j].y;
                    const float denom = tf + nr[j];
                    const float sc = fast_div ? ns_div_short(idf * (tf * (1.2f + 1.0f)), denom) : (idf * (tf * (1.2f + 1.0f))) / denom;
                    x[j] = wq * sc;
                    if (ok[j]) old[j] = vals[ps[j].x - lo];   // all reads of the round first: docIds of one term are distinct
                }
#pragma unroll
                for (int j = 0; j < E; j++) {
                    if (ok[j]) {
                        const uint32_t slot = ps[j].x - lo;
                        vals[slot] = old[j] + x[j];   // -0.0f (untouched) + x == x exactly
                        if (AND) mcnt[slot] = (uint8_t)(mcnt[slot] + 1);
                    }
                }
                s_cur += cnt;
                if (cnt < n) break;   // reached the end of the tile
            }
            if ((uint32_t)lane == t) cur = s_cur;
            wave_sync();   // the next term's read-add-writes follow this term's
        }

        // ---- read the tile back: found, candidates, reset ----
        bool ge_mode = false;   // after a shrink INSIDE this tile, ties with theta may still win on docId
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const float4 q = v4[g * 64 + lane];
            const float vv[4] = {q.x, q.y, q.z, q.w};
            uint32_t cw = 0;
            if (AND) cw = reinterpret_cast<const uint32_t*>(mcnt)[g * 64 + lane];
            bool any = false;
            bool sc_[4];
            bool anyq = false;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                bool touched = __float_as_uint(vv[c]) != kSentinelBits;
                any = any || touched;
                if (AND) touched = touched && (((cw >> (8 * c)) & 0xFFu) == T);   // conjunctive extension
                sc_[c] = touched;
                found_lane += touched ? 1u : 0u;
                anyq = anyq || (touched && vv[c] > theta);
            }
            if (any) {
                v4[g * 64 + lane] = sent4;
                if (AND) reinterpret_cast<uint32_t*>(mcnt)[g * 64 + lane] = 0;
            }
            if (__ballot(anyq) != 0ull) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    bool qf = sc_[c] && (ge_mode ? (vv[c] >= theta) : (vv[c] > theta));
                    unsigned long long mask = __ballot(qf);
                    if (mask != 0ull) {
                        uint32_t n = (uint32_t)__popcll(mask);
                        if (ncand + n > (uint32_t)CB) {
                            ncand = wave_shrink(cand, ncand, theta, K, lane);
                            ge_mode = true;
                            qf = sc_[c] && (vv[c] >= theta);
                            mask = __ballot(qf);
                            n = (uint32_t)__popcll(mask);
                        }
                        if (qf) cand[ncand + lanes_below(mask)] = make_key(vv[c], lo + (uint32_t)((g * 64 + lane) * 4 + c));
                        ncand += n;
                    }
                }
            }
        }
        wave_sync();
        if (ncand > (uint32_t)(CB / 2)) ncand = wave_shrink(cand, ncand, theta, K, lane);
        if (hi >= last_doc) break;
        // nothing left in any list: done
        if (__ballot(((uint32_t)lane < T) && (cur < end)) == 0ull) break;
    }

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

template <int TD, bool AND>
__global__ void __launch_bounds__(256) k_tscore(const DevWItem* __restrict__ items, uint32_t n_items,
                                                const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                                Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                                uint64_t* __restrict__ out_found, uint32_t K) {
    constexpr int WPB = 4;                 // independent waves per workgroup
    __shared__ __attribute__((aligned(16))) float s_vals[WPB][TD];
    __shared__ __attribute__((aligned(16))) uint8_t s_mcnt[WPB][AND ? TD : 16];   // AND: term refs that hit the slot
    __shared__ uint64_t s_cand[WPB][256];
    const int wave = threadIdx.x >> 6;
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
// workgroup — with K > 64 that is the 6th workgroup per CU; 64: the general case, launched only when needed).
// Forcing 72 VGPRs for a 7th workgroup (amdgpu_waves_per_eu) spills and measured no faster.
template <int HK, int FB, bool AND, int CB, int TMAX>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) k_uscore(const DevWItem* __restrict__ items, uint32_t n_items,
                                                const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                                Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                                uint64_t* __restrict__ out_found, uint32_t K) {
    constexpr int WPB = 4;
    __shared__ __attribute__((aligned(16))) uint32_t s_tbl[WPB][2 * HK];                 // HK/2 buckets of 4 entries, or one 2*HK-doc tile
    __shared__ __attribute__((aligned(16))) float s_vals[WPB][FB];                       // driver body: one accumulator per foreign posting
    __shared__ __attribute__((aligned(16))) uint8_t s_mcnt[WPB][AND ? 2 * HK : 16];
    __shared__ uint64_t s_cand[WPB][CB];
    __shared__ __attribute__((aligned(16))) uint4 s_tab[WPB][TMAX];
    __shared__ uint32_t s_aux[WPB][TMAX];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const uint32_t item_idx = blockIdx.x * WPB + wave;
    if (item_idx >= n_items) return;   // whole wave leaves; there is no workgroup barrier in this kernel
    DevWItem it = items[item_idx];
    const bool tiles = (it.whole & 2u) != 0;
    const bool thin = (it.whole & 4u) != 0;   // the non-driver lists are thin: smallest foreign budget
    it.whole &= 9u;   // bit 0: whole segment, bit 3: short division
    if (thin)
        dscore_body<HK / 2, 64, AND, CB>(it, terms, segs, s_tbl[wave], s_vals[wave], s_mcnt[wave], s_cand[wave],
                                 s_tab[wave], s_aux[wave], out_hits, out_nhits, out_found, K, lane);
    else if (tiles)
        tscore_body<2 * HK, AND, CB>(it, terms, segs, reinterpret_cast<float*>(s_tbl[wave]), s_mcnt[wave], s_cand[wave],
                                 out_hits, out_nhits, out_found, K, lane);
    else
        dscore_body<HK / 2, FB, AND, CB>(it, terms, segs, s_tbl[wave], s_vals[wave], s_mcnt[wave], s_cand[wave],
                                 s_tab[wave], s_aux[wave], out_hits, out_nhits, out_found, K, lane);
}

}  // namespace ns

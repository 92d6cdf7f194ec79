// k_wscore — the wave-private scoring kernel (default path for term groups of <= 64 terms).
//
// One WAVE (64 lanes) scores one work item = one (query, segment) term group over a doc range,
// with no workgroup barrier anywhere: every wave of the chip is an independent worker, so posting
// loads of ~20-28 waves per CU overlap and nothing waits on a slower wave.
//
// Per wave, in LDS: an HB-entry accumulator table (fp32 value + docId key), a 256-entry candidate
// buffer (64-bit sort keys).  The wave walks its posting lists in BATCHES of <= HB/2 postings:
//   1. lane t owns term t: cursor, list end, idf, weight.  Window sizes w_t are proportional to the
//      remaining list lengths, so all windows span about the same doc range.
//   2. one probe per term of the last docId in its window; hi = min over terms: every posting with
//      docId <= hi of every term lies inside its window (lists are docId-ascending).
//   3. term by term (query-term order == fp32 accumulation order of src/api_engine.cpp:480), chunk
//      by chunk of 64: coalesced loads of {docId,tf} and of the per-posting norm (no dependent
//      gather), BM25 term score in the reference's operation order, ds_add_f32 into the table —
//      direct-mapped when the batch spans <= HB docs, else open-addressing hash keyed by docId.
//   4. read the table back from registers: `found` (:495), candidates above the running K-th best
//      (:485-492) into the candidate buffer (bitonic-sorted by the wave when it fills), reset.
// The doc ranges of successive batches ascend, which is what makes "score > theta" an exact filter
// under the canonical tie order (score desc, docId asc).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ns_internal.h"

namespace ns {

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in issue order; this only stops the compiler from moving
    // LDS accesses across a phase boundary.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, (uint32_t)__shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// in-LDS bitonic sort (descending) of a[0..P), P a power of two, by ONE wave
__device__ __forceinline__ void wave_bitonic(uint64_t* a, uint32_t P, int lane, bool ascending) {
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t q = lane; q < (P >> 1); q += 64) {
                uint32_t i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                uint32_t p = i | j;
                uint64_t x = a[i], y = a[p];
                bool desc = ((i & k) == 0) != ascending;
                if (desc ? (x < y) : (x > y)) { a[i] = y; a[p] = x; }
            }
            wave_sync();
        }
    }
}
__device__ __forceinline__ void wave_bitonic_desc(uint64_t* a, uint32_t P, int lane) { wave_bitonic(a, P, lane, false); }
// a[0..P) is bitonic (here: a descending run followed by an ascending one): the last log2(P) stages of
// the sorting network leave it sorted descending
__device__ __forceinline__ void wave_bitonic_merge_desc(uint64_t* a, uint32_t P, int lane) {
    for (uint32_t j = P >> 1; j > 0; j >>= 1) {
        for (uint32_t q = lane; q < (P >> 1); q += 64) {
            uint32_t i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
            uint32_t p = i | j;
            uint64_t x = a[i], y = a[p];
            if (x < y) { a[i] = y; a[p] = x; }
        }
        wave_sync();
    }
}

// Sort the wave's candidates, keep the best min(n, K), raise theta to the K-th best.
// Returns (theta bits << 32) | new n  (by value: a reference would push theta to scratch).
__device__ __noinline__ uint64_t wave_shrink_packed(uint64_t* cand, uint32_t n, uint32_t theta_bits, uint32_t K, int lane) {
    uint32_t P = 2;
    while (P < n) P <<= 1;
    for (uint32_t i = n + lane; i < P; i += 64) cand[i] = 0;   // padding sorts last
    wave_sync();
    wave_bitonic_desc(cand, P, lane);
    if (n >= K) {
        theta_bits = __float_as_uint(unorder_bits((uint32_t)(cand[K - 1] >> 32)));   // same address in all lanes: broadcast
        n = K;
    }
    return ((uint64_t)theta_bits << 32) | n;
}
// The same for the 256-entry buffer (K > 32), where the buffer is shrunk every ~90 new candidates.
// `sorted`: the first `sorted` entries are already in descending order (what the previous shrink
// left; new candidates are appended behind them).  With 129..256 entries of which at most 128 are new,
// only the new ones are sorted (128-entry network, ascending) and merged with the old run (8 stages
// over 256): 44 compare-exchange steps per lane instead of 72.
__device__ __noinline__ uint64_t wave_shrink_merge_packed(uint64_t* cand, uint32_t n, uint32_t sorted, uint32_t theta_bits, uint32_t K, int lane) {
    if (sorted > 0 && sorted <= 128u && n > 128u && n - sorted <= 128u) {
        const uint32_t nn = n - sorted;
        uint64_t v0 = 0, v1 = 0;   // 0 = padding: sorts last descending, first ascending
        if ((uint32_t)lane < nn) v0 = cand[sorted + lane];
        if ((uint32_t)(64 + lane) < nn) v1 = cand[sorted + 64 + lane];
        wave_sync();
        for (uint32_t i = sorted + lane; i < 128u; i += 64) cand[i] = 0;
        cand[128 + lane] = v0;
        cand[192 + lane] = v1;
        wave_sync();
        wave_bitonic(cand + 128, 128, lane, true);
        wave_bitonic_merge_desc(cand, 256, lane);
    } else {
        uint32_t P = 2;
        while (P < n) P <<= 1;
        for (uint32_t i = n + lane; i < P; i += 64) cand[i] = 0;   // padding sorts last
        wave_sync();
        wave_bitonic_desc(cand, P, lane);
    }
    if (n >= K) {
        theta_bits = __float_as_uint(unorder_bits((uint32_t)(cand[K - 1] >> 32)));
        n = K;
    }
    return ((uint64_t)theta_bits << 32) | n;
}
__device__ __forceinline__ uint32_t wave_shrink(uint64_t* cand, uint32_t n, float& theta, uint32_t K, int lane) {
    uint64_t r = wave_shrink_packed(cand, n, __float_as_uint(theta), K, lane);
    // wave-uniform by construction; telling the compiler keeps theta, the candidate count and every
    // decision that depends on them in SGPRs (scalar branches instead of exec-masked vector code)
    theta = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(r >> 32)));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r);
}
// CB-aware form used by the scoring bodies: the merge variant (and its `sorted` bookkeeping) exists only in
// the 256-entry instantiations; `sorted` is updated to the new count (a shrink leaves a descending run)
template <int CB>
__device__ __forceinline__ uint32_t wave_shrink_cb(uint64_t* cand, uint32_t n, uint32_t& sorted, float& theta, uint32_t K, int lane) {
    if constexpr (CB > 128) {
        uint64_t r = wave_shrink_merge_packed(cand, n, sorted, __float_as_uint(theta), K, lane);
        theta = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(r >> 32)));
        sorted = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r);
        return sorted;
    } else {
        return wave_shrink(cand, n, theta, K, lane);
    }
}

// ballot straight from the compare (HIP's __ballot goes through an int and costs two extra vector instructions)
__device__ __forceinline__ uint64_t wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// a / b, correctly rounded.  `fast`: the caller guarantees a is +0 or in [2^-32, 2^64] and b in [2^-20, 2^34]
// (checked on the host per item: idf in [2^-30, 2^30], norms in [2^-20, 2^30]); there v_div_scale_f32 scales
// nothing and v_div_fixup_f32 fixes nothing, so the compiler's IEEE division sequence reduces to its core:
// the SAME instructions on the SAME values, i.e. bit-identical results with three instructions less.
__device__ __forceinline__ float ns_div_short(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    float t = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(t, r, q);
}
// Two divisions per lane in the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 are IEEE, full rate on gfx950:
// the same seven roundings per division as ns_div_short, at half the VALU issue slots; only v_rcp_f32 stays scalar).
typedef float ns_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ns_f2 ns_div_short2(ns_f2 a, ns_f2 b) {
    ns_f2 r;
    r.x = __builtin_amdgcn_rcpf(b.x);
    r.y = __builtin_amdgcn_rcpf(b.y);
    const ns_f2 one = {1.0f, 1.0f};
    const ns_f2 e = __builtin_elementwise_fma(-b, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    ns_f2 q = a * r;
    ns_f2 t = __builtin_elementwise_fma(-b, q, a);
    q = __builtin_elementwise_fma(t, r, q);
    t = __builtin_elementwise_fma(-b, q, a);
    return __builtin_elementwise_fma(t, r, q);
}
// q[j] = a[j] / b[j] for N independent lanes-wide divisions; `fast` is wave-uniform (one scalar branch)
template <int N>
__device__ __forceinline__ void ns_div_n(float (&q)[N], const float (&a)[N], const float (&b)[N], bool fast) {
    if (fast) {
#pragma unroll
        for (int j = 0; j + 1 < N; j += 2) {
            const ns_f2 a2 = {a[j], a[j + 1]}, b2 = {b[j], b[j + 1]};
            const ns_f2 q2 = ns_div_short2(a2, b2);
            q[j] = q2.x; q[j + 1] = q2.y;
        }
        if (N & 1) q[N - 1] = ns_div_short(a[N - 1], b[N - 1]);
    } else {
#pragma unroll
        for (int j = 0; j < N; j++) q[j] = a[j] / b[j];
    }
}

__device__ __forceinline__ uint32_t list_lower_bound(const uint2* lst, uint32_t count, uint32_t doc) {
    uint32_t lo = 0, hi = count;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (lst[mid].x < doc) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Both ends of a doc range in one list at once: the two binary searches advance in lockstep, so each
// step has two independent loads in flight instead of one (the searches are pure HBM latency: ~20
// dependent steps per end for a hot list, and every doc-range work item starts with them).
// The first step of each search is an interpolation guess (lists are close to uniform in docId):
// it usually cuts the bracket to a few hundred postings.
__device__ __forceinline__ void list_range(const uint2* lst, uint32_t count, uint32_t doc_lo, uint32_t doc_hi, uint32_t n_docs,
                                           uint32_t& first, uint32_t& last) {
    uint32_t a0 = 0, b0 = count, a1 = 0, b1 = count;
    if (count > 64u && n_docs > 0u) {
        // guess +- a margin, checked by loading both bracket ends (4 independent loads)
        const float dens = (float)count / (float)n_docs;
        const uint32_t margin = 64u + (uint32_t)(8.0f * __builtin_sqrtf((float)count * (1.0f - (dens < 1.0f ? dens : 1.0f)) + 1.0f));
        const uint32_t g0 = min(count, (uint32_t)((float)doc_lo * dens)), g1 = min(count, (uint32_t)((float)doc_hi * dens));
        const uint32_t l0 = g0 > margin ? g0 - margin : 0u, h0 = min(count - 1u, g0 + margin);
        const uint32_t l1 = g1 > margin ? g1 - margin : 0u, h1 = min(count - 1u, g1 + margin);
        const uint32_t vl0 = lst[l0].x, vh0 = lst[h0].x, vl1 = lst[l1].x, vh1 = lst[h1].x;
        // lower_bound(d) lies in (l, h] when lst[l] < d <= lst[h]
        if (vl0 < doc_lo) a0 = l0 + 1u;
        if (vh0 >= doc_lo) b0 = h0;
        if (vl1 < doc_hi) a1 = l1 + 1u;
        if (vh1 >= doc_hi) b1 = h1;
        if (a0 > b0) { a0 = 0; b0 = count; }   // cannot happen for a sorted list
        if (a1 > b1) { a1 = 0; b1 = count; }
    }
    while (a0 < b0 || a1 < b1) {
        const uint32_t m0 = a0 + ((b0 - a0) >> 1), m1 = a1 + ((b1 - a1) >> 1);
        const uint32_t v0 = lst[min(m0, count - 1u)].x, v1 = lst[min(m1, count - 1u)].x;
        if (a0 < b0) { if (v0 < doc_lo) a0 = m0 + 1u; else b0 = m0; }
        if (a1 < b1) { if (v1 < doc_hi) a1 = m1 + 1u; else b1 = m1; }
    }
    first = a0;
    last = a1;
}

// DPP wave reductions (gfx9 row_shr / row_bcast forms): 6 VALU instructions, no LDS round trips.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROWMASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_min_dpp(uint32_t v) {
    v = min(v, dpp_mov<0x111, 0xf>(0xFFFFFFFFu, v));   // row_shr:1
    v = min(v, dpp_mov<0x112, 0xf>(0xFFFFFFFFu, v));   // row_shr:2
    v = min(v, dpp_mov<0x114, 0xf>(0xFFFFFFFFu, v));   // row_shr:4
    v = min(v, dpp_mov<0x118, 0xf>(0xFFFFFFFFu, v));   // row_shr:8
    v = min(v, dpp_mov<0x142, 0xa>(0xFFFFFFFFu, v));   // row_bcast:15
    v = min(v, dpp_mov<0x143, 0xc>(0xFFFFFFFFu, v));   // row_bcast:31
    return rdlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v) {
    v = max(v, dpp_mov<0x111, 0xf>(0u, v));
    v = max(v, dpp_mov<0x112, 0xf>(0u, v));
    v = max(v, dpp_mov<0x114, 0xf>(0u, v));
    v = max(v, dpp_mov<0x118, 0xf>(0u, v));
    v = max(v, dpp_mov<0x142, 0xa>(0u, v));
    v = max(v, dpp_mov<0x143, 0xc>(0u, v));
    return rdlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += dpp_mov<0x111, 0xf>(0u, v);
    v += dpp_mov<0x112, 0xf>(0u, v);
    v += dpp_mov<0x114, 0xf>(0u, v);
    v += dpp_mov<0x118, 0xf>(0u, v);
    v += dpp_mov<0x142, 0xa>(0u, v);
    v += dpp_mov<0x143, 0xc>(0u, v);
    return v;
}

typedef unsigned int nat_u2 __attribute__((ext_vector_type(2)));
// explicit global address space: pointers loaded from a descriptor are generic to the compiler, and
// generic (flat_*) loads also tick lgkmcnt, which would serialise them with the LDS phases
typedef const __attribute__((address_space(1))) nat_u2* gp_u2;
typedef const __attribute__((address_space(1))) float* gp_f32;

// D   direct-mapped accumulator slots per wave (a batch spanning <= D docs needs no hashing)
// HK  hash keys per wave (<= D); a batch holds at most HK/2 postings
template <int D, int HK, bool AND>
__global__ void __launch_bounds__(256) k_wscore(const DevWItem* __restrict__ items, uint32_t n_items,
                                                const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                                Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                                uint64_t* __restrict__ out_found, uint32_t K) {
    constexpr int WPB = 4;                 // independent waves per workgroup
    constexpr int CB = 256;                // candidate buffer entries (>= NS_MAX_K + 64, power of two)
    constexpr int BUDGET = HK / 2;         // postings per batch (hash load factor <= 1/2)
    constexpr int E = BUDGET / 64;         // postings per lane per batch
    constexpr int LOG2HK = (HK == 256) ? 8 : (HK == 512 ? 9 : (HK == 1024 ? 10 : 11));
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    static_assert(HK == 256 || HK == 512 || HK == 1024 || HK == 2048, "HK must be 256..2048");
    static_assert(D >= HK && D % 256 == 0, "D must be a multiple of 256 and >= HK");

    __shared__ __attribute__((aligned(16))) float s_vals[WPB][D];
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[WPB][HK];
    __shared__ __attribute__((aligned(16))) uint8_t s_mcnt[WPB][AND ? D : 16];   // AND: term refs that hit the slot
    __shared__ uint64_t s_cand[WPB][CB];
    __shared__ __attribute__((aligned(16))) uint4 s_tab[WPB][64];   // per term: {idf, qweight, first posting - excl prefix, first posting}
    __shared__ uint32_t s_aux[WPB][64];                              // T > 8: inclusive window prefix; then: new cursors

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: the item is fetched with scalar loads
    const int lane = threadIdx.x & 63;
    const uint32_t item_idx = blockIdx.x * WPB + wave;
    if (item_idx >= n_items) return;   // whole wave leaves; there is no workgroup barrier in this kernel

    float* vals = s_vals[wave];
    uint32_t* keys = s_keys[wave];
    uint8_t* mcnt = s_mcnt[wave];
    uint64_t* cand = s_cand[wave];
    uint4* tab = s_tab[wave];
    uint32_t* aux = s_aux[wave];

    const DevWItem it = items[item_idx];
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
        for (int g = 0; g < D / 256; g++) v4[g * 64 + lane] = sent4;
#pragma unroll
        for (int g = 0; g < HK / 256; g++) k4[g * 64 + lane] = empty4;
        if (AND) {
            uint32_t* m32 = reinterpret_cast<uint32_t*>(mcnt);
#pragma unroll
            for (int g = 0; g < D / 256; g++) m32[g * 64 + lane] = 0;
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
            const uint2* lst = seg.postings + tm.list_off;
            cur = list_lower_bound(lst, tm.count, it.doc_lo);
            end = list_lower_bound(lst, tm.count, it.doc_hi);
            if (end < cur) end = cur;
        }
    }
    // postings still to be consumed by this item (scalar, maintained incrementally)
    uint64_t R = 0;
    {
        const uint32_t r32 = end - cur;
        for (uint32_t t = 0; t < T; t++) R += rdlane(r32, t);
    }

    uint32_t lo = it.doc_lo;
    const uint32_t last_doc = it.doc_hi - 1;   // host guarantees doc_hi > doc_lo and doc_hi <= n_docs
    float theta = -__builtin_inff();
    uint32_t ncand = 0;
    uint32_t found_lane = 0;   // per-lane count of scored docs, reduced once at the end

    // Window sizes proportional to what is left of each list (so all windows span about the same doc
    // range; Sum(w) <= BUDGET + fp slop << the table's spare half) + the probe of each window's last
    // docId.  Planned one batch ahead: the probe flies while the current batch is scored.
    uint32_t w = 0, e = 0xFFFFFFFFu, f = 0xFFFFFFFFu;   // f: first docId left in the list (~0: exhausted)
#define NS_PLAN_WINDOWS()                                                                          \
    {                                                                                              \
        const uint32_t rem_ = end - cur;                                                           \
        const uint32_t nact_ = (uint32_t)__popcll(__ballot(rem_ > 0));                             \
        const float scale_ = (float)(BUDGET - (int)nact_) * __builtin_amdgcn_rcpf((float)R);       \
        uint32_t w_ = 1u + (uint32_t)((float)rem_ * scale_);                                       \
        w_ = (w_ < rem_) ? w_ : rem_;                                                              \
        const bool probe_ = w_ < rem_;   /* false for lanes >= T (rem_ == 0) */                    \
        const uint32_t pi_ = probe_ ? (base + cur + w_ - 1u) : 0u;                                 \
        const uint32_t fi_ = (rem_ > 0) ? (base + cur) : 0u;                                       \
        const nat_u2 pv_ = postings[pi_];   /* unconditional loads of valid indices: no branch */  \
        const nat_u2 fv_ = postings[fi_];                                                          \
        w = w_;                                                                                    \
        e = probe_ ? pv_.x : 0xFFFFFFFFu;                                                          \
        f = (rem_ > 0) ? fv_.x : 0xFFFFFFFFu;                                                      \
    }
    if (R > 0) NS_PLAN_WINDOWS();
    wave_sync();

    while (R > 0) {
        // ---- which terms take part in this batch ----
        // The term with the most postings left drives the batch; its window ends at docId Ed.  A
        // term whose next posting lies beyond Ed has nothing to add to the docs of this batch: it
        // stays out (cursor untouched) and costs nothing.  With one hot list and a few sparse ones
        // many batches are EXCLUSIVE to the hot list.
        {
            const uint32_t remv = end - cur;
            const uint32_t mx = wave_max_dpp(remv);
            const uint32_t dl = (uint32_t)__builtin_ctzll(__ballot(remv == mx));   // mx > 0 here (R > 0)
            const uint32_t Ed = rdlane(e, dl);
            const bool inc = (remv > 0) && ((uint32_t)lane == dl || f <= Ed);
            w = inc ? w : 0u;
            e = inc ? e : 0xFFFFFFFFu;
            if (__popcll(__ballot(inc)) == 1 && Ed <= last_doc) {
                // ---- exclusive batch: every doc of [lo, Ed] is touched by this one term only, so its
                //      score is 0.0f + w*s == w*s exactly (src/api_engine.cpp:480): no table, no merge;
                //      the whole window is consumed. ----
                const uint32_t wd = rdlane(w, dl);
                const uint32_t b0 = rdlane(base, dl) + rdlane(cur, dl);
                const float idf = __uint_as_float(rdlane(idf_bits, dl));
                const float wq = __uint_as_float(rdlane(wq_bits, dl));
                nat_u2 ps[E];
                float nr[E];
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const uint32_t p = (uint32_t)(j * 64 + lane);
                    const uint32_t idx = b0 + ((p < wd) ? p : 0u);
                    ps[j] = postings[idx];
                    nr[j] = pnorm[idx];
                }
                if ((uint32_t)lane == dl) cur += wd;
                R = (R > wd) ? (R - wd) : 0;
                const bool xdone = (R == 0) || (Ed >= last_doc);
                if (!xdone) NS_PLAN_WINDOWS();
                float fin[E];
                bool sc_[E];
                bool anyq = false;
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const uint32_t p = (uint32_t)(j * 64 + lane);
                    const float tf = (float)ps[j].y;
                    const float denom = tf + nr[j];
                    const float sc = (idf * (tf * (1.2f + 1.0f))) / denom;
                    fin[j] = wq * sc;
                    // docId outside [lo, Ed] only for corrupt (unsorted) lists: consumed, not scored
                    sc_[j] = (p < wd) && (ps[j].x >= lo) && (ps[j].x <= Ed);
                    if (AND && T > 1) sc_[j] = false;   // conjunctive extension: one term alone never qualifies
                    found_lane += sc_[j] ? 1u : 0u;
                    anyq = anyq || (sc_[j] && fin[j] > theta);
                }
                if (__ballot(anyq) != 0ull) {
                    bool ge_mode = false;
#pragma unroll
                    for (int j = 0; j < E; j++) {
                        bool qf = sc_[j] && (ge_mode ? (fin[j] >= theta) : (fin[j] > theta));
                        unsigned long long mask = __ballot(qf);
                        if (mask != 0ull) {
                            uint32_t n = (uint32_t)__popcll(mask);
                            if (ncand + n > (uint32_t)CB) {
                                ncand = wave_shrink(cand, ncand, theta, K, lane);
                                ge_mode = true;
                                qf = sc_[j] && (fin[j] >= theta);
                                mask = __ballot(qf);
                                n = (uint32_t)__popcll(mask);
                            }
                            if (qf) cand[ncand + lanes_below(mask)] = make_key(fin[j], ps[j].x);
                            ncand += n;
                        }
                    }
                    wave_sync();
                    if (ncand > (uint32_t)(CB / 2)) ncand = wave_shrink(cand, ncand, theta, K, lane);
                }
                if (xdone) break;
                lo = Ed + 1;
                continue;
            }
        }

        // ---- batch geometry (shared batch: several terms have postings in its doc range) ----
        const uint32_t incl = wave_incl_scan_dpp(w);
        const uint32_t total = rdlane(incl, 63);
        if ((uint32_t)lane < T) tab[lane] = make_uint4(idf_bits, wq_bits, base + cur - (incl - w), base + cur);
        if (T > 8 && (uint32_t)lane < T) aux[lane] = incl;
        uint32_t hi = wave_min_dpp(e);   // every posting with docId <= hi of every term is inside its window
        hi = min(hi, last_doc);
        const bool direct = (hi >= lo) && ((hi - lo) < (uint32_t)D);   // uniform
        wave_sync();

        // ---- flat, coalesced loads of the whole batch (all in flight together, no branches:
        //      lanes beyond `total` re-read the batch's first posting and are masked afterwards) ----
        uint32_t tj[E];
#pragma unroll
        for (int j = 0; j < E; j++) tj[j] = 0;
        if (T > 1) {
            if (T <= 8) {
                for (uint32_t t = 0; t + 1 < T; t++) {
                    const uint32_t sp = rdlane(incl, t);
#pragma unroll
                    for (int j = 0; j < E; j++) tj[j] += ((uint32_t)(j * 64 + lane) >= sp) ? 1u : 0u;
                }
            } else {
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const uint32_t p = min((uint32_t)(j * 64 + lane), total - 1u);
                    uint32_t a = 0, b = T - 1;   // smallest t with incl[t] > p
                    while (a < b) {
                        const uint32_t m = (a + b) >> 1;
                        if (aux[m] > p) b = m; else a = m + 1;
                    }
                    tj[j] = a;
                }
                wave_sync();   // aux is reused below
            }
        }
        nat_u2 pst[E];
        float nrm[E];
        uint32_t pidx[E];
#pragma unroll
        for (int j = 0; j < E; j++) {
            const uint32_t p = (uint32_t)(j * 64 + lane);
            const bool inb = p < total;
            tj[j] = inb ? tj[j] : 0u;
            pidx[j] = tab[tj[j]].z + (inb ? p : 0u);
            pst[j] = postings[pidx[j]];
            nrm[j] = pnorm[pidx[j]];
            pst[j].x = inb ? pst[j].x : 0xFFFFFFFFu;   // docId ~0 is never <= hi
        }

        // ---- how much of each window is consumed (docId <= hi): cursor update ----
        // docIds ascend inside a window, so "taken" is a prefix of it: the first posting of a window
        // that is NOT taken (or nothing, if all are) marks the new cursor.  That lane is unique per
        // term, so it publishes its posting index with a plain LDS store; no per-term ballot loops.
        bool take[E];
        uint32_t batch_consumed;
        {
            if ((uint32_t)lane < T) aux[lane] = base + cur + w;   // default: whole window consumed
            wave_sync();
            unsigned long long prev_last = 1ull;   // "element before the batch" counts as taken
#pragma unroll
            for (int j = 0; j < E; j++) {
                take[j] = pst[j].x <= hi;
                const unsigned long long m = __ballot(take[j]);
                // previous flat element taken?  (bit lane-1 of this chunk's mask, or the last lane of the previous chunk)
                const bool prev_take = (((m << 1) | prev_last) >> lane) & 1ull;
                prev_last = m >> 63;
                const bool first_untaken = ((uint32_t)(j * 64 + lane) < total) && !take[j] &&
                                           (prev_take || pidx[j] == tab[tj[j]].w);
                if (first_untaken) aux[tj[j]] = pidx[j];
            }
            wave_sync();
            uint32_t c = 0;
            if ((uint32_t)lane < T) {
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
            cur += w;
            if (cur > end) cur = end;
            batch_consumed = total;
        }
        R = (R > batch_consumed) ? (R - batch_consumed) : 0;
        const bool done = (R == 0) || (hi >= last_doc);
        // plan the next batch now: its docId probes fly while this batch is scored and read back
        if (!done) NS_PLAN_WINDOWS();

        // ---- BM25 term scores + table slots ----
        float x[E];
        uint32_t slot[E];
        bool ok[E];
        // terms present in the batch (uniform): first element's term .. last element's term
        uint32_t tb_min = 0, tb_max = 0;
        if (T > 1) {
            tb_min = rdlane(tj[0], 0);
            const uint32_t lastp = total - 1u;
            uint32_t tl = 0;
#pragma unroll
            for (int j = 0; j < E; j++)
                if ((lastp >> 6) == (uint32_t)j) tl = rdlane(tj[j], lastp & 63u);   // uniform
            tb_max = tl;
        }
#pragma unroll
        for (int j = 0; j < E; j++) {
            const uint4 ent = tab[tj[j]];
            // docId < lo only for corrupt (unsorted) lists: such postings are consumed, not scored
            ok[j] = take[j] && (pst[j].x >= lo);
            // src/api_engine.cpp:477-480, operation for operation (k1 + 1.0f == 0x400CCCCD)
            const float tf = (float)pst[j].y;
            const float denom = tf + nrm[j];
            const float sc = (__uint_as_float(ent.x) * (tf * (1.2f + 1.0f))) / denom;
            x[j] = __uint_as_float(ent.y) * sc;
            slot[j] = pst[j].x - lo;
        }
        if (!direct) {
            // Open-addressing claim WITHOUT LDS atomics (integer and float LDS atomics are serialised
            // per lane on gfx950): read the key; if the slot is free store our docId and read it back —
            // the wave's LDS operations execute in order, so exactly one of the colliding docIds
            // survives the store and everybody else moves on.  Equal docIds (same doc, two terms)
            // agree on the slot.
#pragma unroll
            for (int j = 0; j < E; j++) {
                if ((uint32_t)(j * 64) >= total) continue;   // uniform
                const uint32_t doc = pst[j].x;
                uint32_t sl = (doc * 2654435761u) >> (32 - LOG2HK);
                bool pending = ok[j];
                for (int round = 0; round < HK; round++) {
                    if (__ballot(pending) == 0ull) break;
                    uint32_t k = EMPTY;
                    if (pending) k = keys[sl];
                    if (pending && k == EMPTY) keys[sl] = doc;
                    wave_sync();
                    if (pending && k == EMPTY) k = keys[sl];
                    if (pending && k == doc) pending = false;
                    if (pending) sl = (sl + 1) & (HK - 1);
                }
                slot[j] = sl;
            }
        }

        // ---- accumulate: one read-add-write per term, terms in query order (the fp32 order of
        //      src/api_engine.cpp:480).  docIds are unique inside a term, so the plain RMW is race-free;
        //      reading the sentinel back tells the FIRST posting of a doc that it owns the slot. ----
        bool owner[E];
#pragma unroll
        for (int j = 0; j < E; j++) owner[j] = false;
        for (uint32_t tt = tb_min; tt <= tb_max; tt++) {
            float old[E];
#pragma unroll
            for (int j = 0; j < E; j++) {
                old[j] = 0.0f;
                if (ok[j] && tj[j] == tt) old[j] = vals[slot[j]];
            }
#pragma unroll
            for (int j = 0; j < E; j++) {
                if (ok[j] && tj[j] == tt) {
                    owner[j] = owner[j] || (__float_as_uint(old[j]) == kSentinelBits);
                    vals[slot[j]] = old[j] + x[j];
                    if (AND) mcnt[slot[j]] = (uint8_t)(mcnt[slot[j]] + 1);
                }
            }
            wave_sync();
        }

        // ---- read back through the owners: found (:495), candidates above theta (:485-492), reset ----
        float fin[E];
        bool scored[E];
        bool anyq = false;
#pragma unroll
        for (int j = 0; j < E; j++) {
            fin[j] = 0.0f;
            scored[j] = owner[j];
            if (owner[j]) {
                fin[j] = vals[slot[j]];
                if (AND) scored[j] = (mcnt[slot[j]] == (uint8_t)T);   // conjunctive extension: every term ref hit the doc
            }
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < E; j++) {
            found_lane += scored[j] ? 1u : 0u;
            anyq = anyq || (scored[j] && fin[j] > theta);
            if (owner[j]) {   // the first toucher resets the slot for the next batch
                vals[slot[j]] = __uint_as_float(kSentinelBits);
                if (!direct) keys[slot[j]] = EMPTY;
                if (AND) mcnt[slot[j]] = 0;
            }
        }
        if (__ballot(anyq) != 0ull) {
            bool ge_mode = false;   // after a shrink INSIDE this batch, ties with theta may still win on docId
#pragma unroll
            for (int j = 0; j < E; j++) {
                bool qf = scored[j] && (ge_mode ? (fin[j] >= theta) : (fin[j] > theta));
                unsigned long long mask = __ballot(qf);
                if (mask != 0ull) {
                    uint32_t n = (uint32_t)__popcll(mask);
                    if (ncand + n > (uint32_t)CB) {
                        ncand = wave_shrink(cand, ncand, theta, K, lane);
                        ge_mode = true;
                        qf = scored[j] && (fin[j] >= theta);
                        mask = __ballot(qf);
                        n = (uint32_t)__popcll(mask);
                    }
                    if (qf) cand[ncand + lanes_below(mask)] = make_key(fin[j], pst[j].x);
                    ncand += n;
                }
            }
            wave_sync();
            if (ncand > (uint32_t)(CB / 2)) ncand = wave_shrink(cand, ncand, theta, K, lane);
        }
        if (done) break;
        lo = hi + 1;
    }
#undef NS_PLAN_WINDOWS

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

// per-posting norm, built once at upload: pnorm[i] = norm[postings[i].docId]
__global__ void k_pnorm(const uint2* __restrict__ postings, const float* __restrict__ norm, float* __restrict__ pnorm,
                        uint64_t n_postings, uint32_t n_docs) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_postings; i += stride) {
        uint32_t d = postings[i].x;
        pnorm[i] = d < n_docs ? norm[d] : 1.0f;
    }
}

}  // namespace ns

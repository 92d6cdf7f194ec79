// Wave-level primitives shared by the scoring bodies (ns_driver_kernel.hip, ns_tile_kernel.hip): LDS ordering
// fence, DPP reductions and scans, ballot helpers, the wave bitonic sort / merge of the candidate buffer, the exact
// short form of the BM25 division, list range searches, address-space-qualified pointer types, and k_pnorm.
// (The wave-private batch kernel k_wscore that used to live here — round 1's second design — was retired in
// round 2: the driver-stream and doc-tile bodies superseded it.)
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
// ---- the candidate buffer sorted in REGISTERS ----
// The in-LDS network above pays an LDS round trip per stage (72 stages over 256 entries: the shrink of a K = 100 batch
// took a third of its doc-tile items' time).  Here lane l holds entries l*R .. l*R + R - 1 (R = CB / 64): the stages with
// a partner distance below R are compare-exchanges between registers of one lane, the others exchange with lane l ^ m
// (quad DPP for m = 1, 2; the LDS crossbar without touching memory — ds_swizzle / ds_bpermute — for m = 4 .. 32).
// Keys are unique except for the zero padding, so every correct network yields the same order as the one above.
template <int M>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v, int lane) {
    if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    else if constexpr (M < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (M << 10) | 0x1F);          // bit mode: xor M inside 32 lanes
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, (int)v);
}
template <int R, int KK, int J>
__device__ __forceinline__ void wave_sort_stage(uint64_t (&k)[R], int lane) {
    // entry index i = lane * R + r; stage (KK, J): partner i ^ J, descending where (i & KK) == 0
    if constexpr (J < R) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (r & J) continue;
            const uint64_t a = k[r], b = k[r | J];
            bool desc;
            if constexpr (KK < R) desc = (r & KK) == 0;
            else desc = ((lane * R) & KK) == 0;
            const bool sw = desc ? (a < b) : (a > b);
            k[r] = sw ? b : a;
            k[r | J] = sw ? a : b;
        }
    } else {
        constexpr int M = J / R;
        const bool keep_max = ((lane & M) == 0) == (((lane * R) & KK) == 0);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint64_t x = k[r];
            const uint64_t y = ((uint64_t)lane_xor_u32<M>((uint32_t)(x >> 32), lane) << 32) | lane_xor_u32<M>((uint32_t)x, lane);
            const bool take = keep_max ? (y > x) : (y < x);
            k[r] = take ? y : x;
        }
    }
}
template <int R, int KK, int J>
__device__ __forceinline__ void wave_sort_merge_steps(uint64_t (&k)[R], int lane) {
    wave_sort_stage<R, KK, J>(k, lane);
    if constexpr (J > 1) wave_sort_merge_steps<R, KK, J / 2>(k, lane);
}
template <int R, int KK>
__device__ __forceinline__ void wave_sort_levels(uint64_t (&k)[R], int lane) {
    if constexpr (KK > 2) wave_sort_levels<R, KK / 2>(k, lane);
    wave_sort_merge_steps<R, KK, KK / 2>(k, lane);
}
// cand[0..n) -> sorted descending in place (entries n .. 64 R - 1 come back as zero padding), by one wave
template <int R>
__device__ __forceinline__ void wave_sort_desc_regs(uint64_t* cand, uint32_t n, int lane) {
    uint64_t k[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = (uint32_t)(lane * R + r);
        k[r] = i < n ? cand[i] : 0ull;
    }
    wave_sort_levels<R, 64 * R>(k, lane);
    wave_sync();
#pragma unroll
    for (int r = 0; r < R; r++) cand[lane * R + r] = k[r];
    wave_sync();
}
template <int CB>
__device__ __noinline__ uint64_t wave_shrink_regs_packed(uint64_t* cand, uint32_t n, uint32_t theta_bits, uint32_t K, int lane) {
    wave_sync();
    wave_sort_desc_regs<CB / 64>(cand, n, lane);
    if (n >= K) {
        theta_bits = __float_as_uint(unorder_bits((uint32_t)(cand[K - 1] >> 32)));   // same address in all lanes: broadcast
        n = K;
    }
    return ((uint64_t)theta_bits << 32) | n;
}

template <int CB>
__device__ __forceinline__ uint32_t wave_shrink_cb(uint64_t* cand, uint32_t n, uint32_t& sorted, float& theta, uint32_t K, int lane) {
    static_assert(CB == 128 || CB == 256, "candidate buffer of 128 or 256 entries");
    const uint64_t r = wave_shrink_regs_packed<CB>(cand, n, __float_as_uint(theta), K, lane);
    // wave-uniform by construction; telling the compiler keeps theta, the candidate count and every
    // decision that depends on them in SGPRs (scalar branches instead of exec-masked vector code)
    theta = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(r >> 32)));
    sorted = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r);
    return sorted;
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
typedef const __attribute__((address_space(1))) uint32_t* gp_u32;

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

// Packed posting stream, built once per segment (ns_segment_build_packed): one wave per block of 256 postings.
// nidx[doc] = index of the doc's norm in the segment's table of distinct norms (k_norm over the distinct doc lengths:
// the same expression, the same bits).  A block's doc width follows from the span of its docIds; blocks that straddle
// two lists (docIds not ascending) simply get the width their span needs.
__global__ void __launch_bounds__(64) k_pack(const uint2* __restrict__ postings, const uint16_t* __restrict__ nidx,
                                             uint32_t* __restrict__ packed, uint2* __restrict__ hdr, uint64_t n_postings,
                                             uint32_t n_docs, uint32_t n_blocks) {
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const uint32_t lane = threadIdx.x;
    uint32_t doc[4], tf[4], ni[4];
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint64_t g = (uint64_t)b * kPkBlock + (uint32_t)c * 64u + lane;
        doc[c] = 0xFFFFFFFFu; tf[c] = 0; ni[c] = 0;
        if (g < n_postings) {
            const uint2 p = postings[g];
            doc[c] = p.x; tf[c] = p.y;
            ni[c] = p.x < n_docs ? nidx[p.x] : 0u;
            mn = min(mn, p.x); mx = max(mx, p.x);
        }
    }
    mn = wave_min_dpp(mn);
    mx = wave_max_dpp(mx);
    if (mn > mx) { mn = 0; mx = 0; }   // no posting at all (cannot happen for b < n_blocks)
    const uint32_t span = mx - mn;
    const uint32_t code = span < 256u ? 0u : (span < 65536u ? 1u : 2u);
    const uint32_t base = code == 2u ? 0u : mn;
    uint32_t* blk = packed + (uint64_t)b * kPkStrideDwords;
    uint32_t off[4];
#pragma unroll
    for (int c = 0; c < 4; c++) off[c] = doc[c] == 0xFFFFFFFFu ? 0u : doc[c] - base;   // padding lanes are masked by posting index, never by value
    blk[kPkTf + lane] = min(tf[0], 255u) | (min(tf[1], 255u) << 8) | (min(tf[2], 255u) << 16) | (min(tf[3], 255u) << 24);
    blk[kPkNormA + lane] = ni[0] | (ni[1] << 16);
    blk[kPkNormB + lane] = ni[2] | (ni[3] << 16);
    if (code == 0u) {
        blk[kPkDoc + lane] = off[0] | (off[1] << 8) | (off[2] << 16) | (off[3] << 24);
    } else if (code == 1u) {
        blk[kPkDoc + lane] = off[0] | (off[1] << 16);
        blk[kPkDoc + 64 + lane] = off[2] | (off[3] << 16);
    } else {
#pragma unroll
        for (int c = 0; c < 4; c++) blk[kPkDoc + c * 64 + lane] = doc[c];
    }
    if (lane == 0) hdr[b] = make_uint2(base, code);
}

}  // namespace ns

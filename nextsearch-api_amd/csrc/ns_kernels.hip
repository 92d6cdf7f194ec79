// Hand-written CDNA4 (gfx950, wave64) kernels for NextSearch's posting-traversal / BM25 / top-k
// hot path — the device side of cord19::Engine::search, /root/reference/src/api_engine.cpp:441-504.
//
// This is HBM/L2-bound integer + fp32-scalar work: no MFMA.  What matters here is coalesced
// posting reads, LDS-resident accumulation, few barriers and enough workgroups per CU.
//
//   k_norm    per-doc BM25 length norm, once per segment upload           (api_engine.cpp:478, doc part)
//   k_bounds  per (query,segment) term group: posting index of every doc-tile boundary in every
//             posting list (binary search) -> [tile][term] table
//   k_score   one workgroup per work item (term group x doc-tile range):
//               for each tile: flat, coalesced read of the tile's sub-lists of ALL terms at once,
//               fp32 BM25 term score (exact reference operation order, contraction off),
//               ds_add_f32 into a dense LDS accumulator tile in query-term order (:473-481),
//               register scan of the tile -> `found` (:495) + candidates above the running k-th
//               best -> LDS candidate buffer, bitonic-sorted when it fills (:485-492)
//             -> <=K hits sorted by (score desc, doc asc) + found per work item
//   k_merge   per query: K-way tournament over its work items' sorted partial lists (global heap
//             across segments, :434-435,:499-504) + sum of found
//
// Bit-exactness: compiled with -ffp-contract=off; division is the IEEE-correct v_div_* sequence
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); idf comes from the host (glibc logf).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ns_internal.h"

namespace ns {

// ------------------------------------------------------------------------------------------------
__global__ void k_norm(const uint32_t* __restrict__ doc_len, float* __restrict__ norm, uint32_t n, float avgdl) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float dl = (float)doc_len[i];
        // k1 * (1.0f - b + b * (dl / avgdl)), k1 = 1.2f, b = 0.75f  (src/api_engine.cpp:375-376,478)
        norm[i] = 1.2f * ((1.0f - 0.75f) + 0.75f * (dl / avgdl));
    }
}

// ------------------------------------------------------------------------------------------------
// bounds[g.bounds_off + tile*T + t] = first posting index i of term t with docId >= min(tile*tile_docs, N)
__global__ void __launch_bounds__(128) k_bounds(const DevGroup* __restrict__ groups, const DevTerm* __restrict__ terms,
                                                const DevSeg* __restrict__ segs, uint32_t* __restrict__ bounds,
                                                uint32_t tile_docs) {
    const DevGroup g = groups[blockIdx.x];
    const DevSeg seg = segs[g.seg];
    const uint32_t T = g.term_count;
    const uint32_t total = (seg.n_tiles + 1) * T;
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        uint32_t tile = e / T, t = e - tile * T;
        const DevTerm term = terms[g.term_begin + t];
        uint64_t dlim64 = (uint64_t)tile * tile_docs;
        uint32_t dlim = dlim64 > seg.n_docs ? seg.n_docs : (uint32_t)dlim64;
        const uint2* lst = seg.postings + term.list_off;
        uint32_t lo = 0, hi = term.count;
        while (lo < hi) {
            uint32_t mid = lo + ((hi - lo) >> 1);
            if (lst[mid].x < dlim) lo = mid + 1; else hi = mid;
        }
        bounds[g.bounds_off + e] = lo;
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t order_bits(float f) {   // monotone float -> u32
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorder_bits(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}
// larger key == better hit: higher score, then smaller docId
__device__ __forceinline__ uint64_t make_key(float s, uint32_t doc) {
    return ((uint64_t)order_bits(s) << 32) | (uint64_t)(0xFFFFFFFFu - doc);
}

template <int NT>
__device__ __forceinline__ void wg_bitonic_desc(uint64_t* a, uint32_t P) {
    // in-LDS bitonic sort of a[0..P), P a power of two, descending; all NT threads participate
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t q = threadIdx.x; q < (P >> 1); q += NT) {
                uint32_t i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                uint32_t p = i | j;
                uint64_t x = a[i], y = a[p];
                bool desc = (i & k) == 0;
                if (desc ? (x < y) : (x > y)) { a[i] = y; a[p] = x; }
            }
            __syncthreads();
        }
    }
}

// Sort the candidate buffer, keep the best min(n, K), raise theta to the K-th best.
// Must be called by all threads with all prior candidate writes visible (after a barrier).
// Returns the new candidate count, computed identically by every thread (callers must not re-read
// *s_cnt next to code that appends: a fast thread's append would be seen by a slow thread's read).
template <int NT, int CAP>
__device__ __noinline__ uint32_t wg_shrink(uint64_t* cand, uint32_t* s_cnt, float* s_theta, uint32_t K) {
    const uint32_t n = *s_cnt;
    uint32_t P = 2;
    while (P < n) P <<= 1;
    for (uint32_t i = n + threadIdx.x; i < P; i += NT) cand[i] = 0;   // padding sorts last
    __syncthreads();
    wg_bitonic_desc<NT>(cand, P);
    if (threadIdx.x == 0) {
        if (n >= K) {
            *s_cnt = K;
            *s_theta = unorder_bits((uint32_t)(cand[K - 1] >> 32));
        }
    }
    __syncthreads();
    return n >= K ? K : n;
}

// NT   threads per workgroup (multiple of 64)
// SPT  accumulator slots per thread (multiple of 4)  -> tile = NT*SPT docs, NT*SPT*4 bytes of LDS
// U    postings per thread per round
// AND  conjunctive extension: per-slot match counters
template <int NT, int SPT, int U, bool AND>
__global__ void __launch_bounds__(NT) k_score(const DevItem* __restrict__ items, const DevTerm* __restrict__ terms,
                                              const DevSeg* __restrict__ segs, const uint32_t* __restrict__ bounds,
                                              Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                              uint64_t* __restrict__ out_found, uint32_t K) {
    constexpr int TILE = NT * SPT;
    constexpr int NG = SPT / 4;                       // float4 groups per thread
    constexpr int CAP = (NT >= 1024) ? 2048 : (NT >= 512 ? 1024 : 512);   // >= NT + NS_MAX_K, power of two
    constexpr int TG = 64;                            // terms handled per pass (one wave builds the prefix)
    static_assert(SPT % 4 == 0, "SPT must be a multiple of 4");
    static_assert(CAP >= NT + 100, "candidate buffer too small");

    __shared__ __attribute__((aligned(16))) float acc[TILE];
    __shared__ __attribute__((aligned(16))) uint32_t mcnt[AND ? TILE / 4 : 4];   // packed u8 match counters
    __shared__ uint64_t cand[CAP];
    __shared__ uint64_t s_pbase[TG];      // first posting (absolute index) of each term's sub-list in this tile
    __shared__ uint32_t s_pref[TG + 1];   // exclusive prefix of sub-list lengths
    __shared__ float s_idf[TG], s_w[TG];
    __shared__ uint32_t s_cnt, s_qual, s_found;
    __shared__ float s_theta;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const DevItem it = items[blockIdx.x];
    const DevSeg seg = segs[it.seg];
    const uint32_t T = it.term_count;
    const uint2* __restrict__ postings = seg.postings;
    const float* __restrict__ norm = seg.norm;

    // ---- init LDS ----
    {
        float4 sent4 = make_float4(__uint_as_float(kTileEmptyBits), __uint_as_float(kTileEmptyBits),
                                   __uint_as_float(kTileEmptyBits), __uint_as_float(kTileEmptyBits));
        float4* acc4 = reinterpret_cast<float4*>(acc);
#pragma unroll
        for (int g = 0; g < NG; g++) acc4[g * NT + tid] = sent4;
        if (AND) {
#pragma unroll
            for (int g = 0; g < NG; g++) mcnt[g * NT + tid] = 0;
        }
        if (tid == 0) { s_cnt = 0; s_qual = 0; s_found = 0; s_theta = -__builtin_inff(); }
    }
    __syncthreads();

    const uint32_t required = T;   // AND: every term ref of the group must hit the doc

    bool skipped_prev = false;
    for (uint32_t tile = it.tile_begin; tile < it.tile_end; tile++) {
        const uint32_t lo = tile * (uint32_t)TILE;
        const uint32_t span = min((uint32_t)TILE, seg.n_docs - lo);
        bool any_work = false;
        // a skipped tile ends without a barrier: slower waves may still be reading s_pref
        if (skipped_prev) __syncthreads();

        for (uint32_t tg = 0; tg < T; tg += TG) {
            const uint32_t ng = min((uint32_t)TG, T - tg);
            // ---- sub-list ranges of this tile for terms tg..tg+ng (wave 0) ----
            if (tid < 64) {
                uint32_t len = 0;
                if ((uint32_t)tid < ng) {
                    const DevTerm tm = terms[it.term_begin + tg + tid];
                    const uint32_t* brow = bounds + it.bounds_off + (uint64_t)tile * T + tg + tid;
                    uint32_t b0 = brow[0], b1 = brow[T];
                    len = b1 - b0;
                    s_pbase[tid] = tm.list_off + b0;
                    s_idf[tid] = tm.idf;
                    s_w[tid] = tm.weight;
                }
                uint32_t incl = len;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    uint32_t v = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += v;
                }
                s_pref[tid] = incl - len;
                if (tid == 63) s_pref[64] = incl;
            }
            __syncthreads();
            const uint32_t L = s_pref[TG];

            for (uint32_t base = 0; base < L; base += NT * U) {
                any_work = true;
                uint32_t slot[U];
                float x[U];
                uint32_t tj[U];
                uint2 pst[U];
                bool valid[U];
                // term of this thread's first element: binary search over the (<=64-entry) prefix
                uint32_t t = 0;
                {
                    uint32_t p0 = base + tid;
                    uint32_t a = 0, b = ng;    // find largest t with s_pref[t] <= p0 (t < ng)
                    while (b - a > 1) {
                        uint32_t m = (a + b) >> 1;
                        if (s_pref[m] <= p0) a = m; else b = m;
                    }
                    t = a;
                }
#pragma unroll
                for (int j = 0; j < U; j++) {
                    uint32_t p = base + j * NT + tid;
                    valid[j] = p < L;
                    tj[j] = 0;
                    pst[j] = make_uint2(0, 0);
                    if (valid[j]) {
                        while (p >= s_pref[t + 1]) t++;   // skips empty sub-lists too
                        tj[j] = t;
                        pst[j] = postings[s_pbase[t] + (p - s_pref[t])];
                    }
                }
                float nrm[U];
#pragma unroll
                for (int j = 0; j < U; j++) {
                    uint32_t d = pst[j].x - lo;
                    valid[j] = valid[j] && (d < span);   // guards corrupt (unsorted / out-of-range) lists
                    slot[j] = d;
                    nrm[j] = valid[j] ? norm[pst[j].x] : 1.0f;
                }
#pragma unroll
                for (int j = 0; j < U; j++) {
                    // src/api_engine.cpp:477-480, operation for operation (k1 + 1.0f == 0x400CCCCD)
                    float tf = (float)pst[j].y;
                    float denom = tf + nrm[j];
                    float s = (s_idf[tj[j]] * (tf * (1.2f + 1.0f))) / denom;
                    x[j] = s_w[tj[j]] * s;
                }
                // ---- ordered accumulation: term t's adds happen after all adds of terms < t (the fp32 order of
                // src/api_engine.cpp:449,480): a barrier separates the terms, and inside one term every posting has
                // a doc of its own, so a plain read-add-write is race-free — no LDS float atomics (ds_add_f32 costs
                // ~3 clk per lane on gfx950, and an atomic could not start an untouched slot from +0.0f).
                uint32_t t_first, t_last;
                {
                    uint32_t pe = min(base + (uint32_t)(NT * U), L) - 1;
                    uint32_t a = 0, b = ng;
                    while (b - a > 1) { uint32_t m = (a + b) >> 1; if (s_pref[m] <= base) a = m; else b = m; }
                    t_first = a;
                    a = 0; b = ng;
                    while (b - a > 1) { uint32_t m = (a + b) >> 1; if (s_pref[m] <= pe) a = m; else b = m; }
                    t_last = a;
                }
                for (uint32_t tt = t_first; tt <= t_last; tt++) {
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        if (valid[j] && tj[j] == tt) {
                            const float old = acc[slot[j]];
                            acc[slot[j]] = ((__float_as_uint(old) == kTileEmptyBits) ? 0.0f : old) + x[j];   // an untouched slot starts at the reference's +0.0f
                            if (AND) atomicAdd(&mcnt[slot[j] >> 2], 1u << ((slot[j] & 3) * 8));
                        }
                    }
                }
            }
            if (tg + TG < T) __syncthreads();   // s_pref/s_pbase are rewritten by the next term pass
        }
        skipped_prev = !any_work;
        if (!any_work) continue;   // uniform: no posting of any term falls into this tile
        __syncthreads();

        // ---- scan the tile from registers: found, candidates above theta, reset ----
        float v[SPT];
        uint32_t tmask = 0;   // touched (and, for AND, fully matched) slots of this thread
        {
            const float4* acc4 = reinterpret_cast<const float4*>(acc);
#pragma unroll
            for (int g = 0; g < NG; g++) {
                float4 q = acc4[g * NT + tid];
                v[4 * g + 0] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
            }
        }
        uint32_t rmask = 0;   // slots to reset (touched at all)
#pragma unroll
        for (int j = 0; j < SPT; j++)
            if (__float_as_uint(v[j]) != kTileEmptyBits) rmask |= 1u << j;
        tmask = rmask;
        if (AND) {
#pragma unroll
            for (int g = 0; g < NG; g++) {
                uint32_t w = mcnt[g * NT + tid];
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (((w >> (8 * c)) & 0xFFu) != required) tmask &= ~(1u << (4 * g + c));
            }
        }
        const float theta = s_theta;
        // s_cnt only changes in the append phase below and in wg_shrink: snapshot it here, where it
        // is stable, NOT after the barrier (another thread may already be appending by then).
        const uint32_t cnt0 = s_cnt;
        uint32_t qmask = 0;
#pragma unroll
        for (int j = 0; j < SPT; j++)
            if (((tmask >> j) & 1u) && v[j] > theta) qmask |= 1u << j;
        {
            // wave totals via ballot-free reduction: popcounts summed with DPP-style shuffles
            uint32_t nt = __popc(tmask), nq = __popc(qmask);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                nt += __shfl_xor(nt, d, 64);
                nq += __shfl_xor(nq, d, 64);
            }
            if (lane == 0) {
                if (nt) atomicAdd(&s_found, nt);
                if (nq) atomicAdd(&s_qual, nq);
            }
        }
        __syncthreads();
        const uint32_t nqual = s_qual;   // complete after the barrier; reset only after the end-of-tile barrier
        if (cnt0 + nqual <= (uint32_t)CAP) {
            // fast path: everything above theta fits
            if (qmask) {
                uint32_t pos = atomicAdd(&s_cnt, (uint32_t)__popc(qmask));
#pragma unroll
                for (int j = 0; j < SPT; j++) {
                    if ((qmask >> j) & 1u) {
                        uint32_t doc = lo + (uint32_t)(((j >> 2) * NT + tid) * 4 + (j & 3));
                        cand[pos++] = make_key(v[j], doc);
                    }
                }
            }
        } else {
            // slow path (first tiles of an item, or adversarial score order): feed the tile in
            // sub-batches of NT slots, shrinking whenever the next sub-batch might not fit.
            // ">=" (not ">") because a shrink inside the tile may have set theta from docs with
            // larger ids than later sub-batches hold; the sort's docId tie-break settles those.
            for (int sb = 0; sb < SPT; sb++) {
                __syncthreads();                 // appends of the previous sub-batch are complete
                const uint32_t c = s_cnt;
                __syncthreads();                 // everyone has read c before anyone appends again
                if (c + (uint32_t)NT > (uint32_t)CAP) wg_shrink<NT, CAP>(cand, &s_cnt, &s_theta, K);
                const float th = s_theta;
                uint32_t sl = (uint32_t)sb * NT + tid;
                float sv = acc[sl];
                bool ok = __float_as_uint(sv) != kTileEmptyBits;
                if (AND) ok = ok && (((mcnt[sl >> 2] >> (8 * (sl & 3))) & 0xFFu) == required);
                if (ok && sv >= th) {
                    uint32_t pos = atomicAdd(&s_cnt, 1u);
                    cand[pos] = make_key(sv, lo + sl);
                }
            }
        }
        // reset touched slots for the next tile
        {
            float4 sent4 = make_float4(__uint_as_float(kTileEmptyBits), __uint_as_float(kTileEmptyBits),
                                       __uint_as_float(kTileEmptyBits), __uint_as_float(kTileEmptyBits));
            float4* acc4 = reinterpret_cast<float4*>(acc);
#pragma unroll
            for (int g = 0; g < NG; g++) {
                if ((rmask >> (4 * g)) & 0xFu) {
                    acc4[g * NT + tid] = sent4;
                    if (AND) mcnt[g * NT + tid] = 0;
                }
            }
        }
        __syncthreads();
        if (tid == 0) s_qual = 0;
        // uniform: nothing appends between this barrier and the next tile's append phase
        if (s_cnt > (uint32_t)(CAP / 2)) wg_shrink<NT, CAP>(cand, &s_cnt, &s_theta, K);
    }

    // ---- final selection for this work item ----
    __syncthreads();
    wg_shrink<NT, CAP>(cand, &s_cnt, &s_theta, K);
    const uint32_t n = min(s_cnt, K);
    Hit* oh = out_hits + (uint64_t)it.out_slot * K;
    for (uint32_t i = tid; i < K; i += NT) {
        Hit h;
        if (i < n) {
            uint64_t key = cand[i];
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
    if (tid == 0) {
        out_nhits[it.out_slot] = n;
        out_found[it.out_slot] = (uint64_t)s_found;
    }
}

// ------------------------------------------------------------------------------------------------
// Joining the partial rows of one query.  Canonical order: score desc, seg asc, doc asc (the reference
// leaves ties unspecified).
//
// merge_rows_wave: one wave, K rounds of "every lane proposes the best head among its rows, the wave
// picks the winner" — O(K * rows / 64) row visits, right for the few rows a query of a large batch has.
// k_merge_wide: one workgroup per query, for queries cut into MANY rows (a lone query is spread over
// the whole chip: thousands of rows).  Rows are sorted, so the answer lies in the rows' prefixes with
// score >= theta, where theta is (a lower bound of) the K-th largest row HEAD: two 11-bit histogram
// passes over the heads find it, the prefixes are gathered into LDS, sorted once, and the first K
// leave.  Cost is independent of K and linear in rows / 256.
__host__ __device__ inline bool merge_is_wide(uint32_t part_count, uint32_t K) { return part_count > 64 && part_count >= K; }

constexpr int kMergeStage = 2048;   // entries k_merge stages in LDS per query
__device__ __forceinline__ uint32_t merge_wave_max(uint32_t v) {   // DPP reduction (row_shr / row_bcast forms), wave-uniform result
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ void merge_rows_wave(uint32_t q, uint32_t pb, uint32_t pc, const Hit* __restrict__ part_hits,
                                                const uint32_t* __restrict__ part_nhits, Hit* __restrict__ out_hits,
                                                uint32_t* __restrict__ out_nhits, uint64_t* __restrict__ out_found, uint64_t found,
                                                uint32_t K, uint32_t* __restrict__ heads, int lane) {
    for (uint32_t i = lane; i < pc; i += 64) heads[pb + i] = 0;
    Hit* oh = out_hits + (uint64_t)q * K;
    uint32_t produced = 0;
    for (; produced < K; produced++) {
        // each lane proposes the best head among its rows
        uint32_t best_s = 0;                 // order_bits(score); 0 == nothing
        uint64_t best_id = ~0ull;            // (seg << 32) | doc, smaller is better
        uint32_t best_row = 0xFFFFFFFFu;
        for (uint32_t i = lane; i < pc; i += 64) {
            uint32_t h = heads[pb + i];
            if (h < part_nhits[pb + i]) {
                Hit e = part_hits[(uint64_t)(pb + i) * K + h];
                uint32_t s = order_bits(e.score);
                uint64_t id = ((uint64_t)e.seg << 32) | e.doc;
                if (best_row == 0xFFFFFFFFu || s > best_s || (s == best_s && id < best_id)) {
                    best_s = s; best_id = id; best_row = pb + i;
                }
            }
        }
        // wave argmax on (best_s desc, best_id asc)
        uint32_t ws = (best_row == 0xFFFFFFFFu) ? 0u : best_s;
        bool has = best_row != 0xFFFFFFFFu;
        uint32_t ms = ws;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) ms = max(ms, (uint32_t)__shfl_xor(ms, d, 64));
        unsigned long long anyb = __ballot(has);
        if (anyb == 0ull) break;
        uint64_t cid = (has && ws == ms) ? best_id : ~0ull;
        uint64_t mid = cid;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            uint64_t o = __shfl_xor(mid, d, 64);
            mid = o < mid ? o : mid;
        }
        if (has && ws == ms && best_id == mid) {
            // unique winner ((seg,doc) pairs are unique across rows)
            Hit h;
            h.score = unorder_bits(ms);
            h.seg = (uint32_t)(mid >> 32);
            h.doc = (uint32_t)mid;
            oh[produced] = h;
            heads[best_row] += 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t i = produced + lane; i < K; i += 64) {
        Hit h;
        h.score = -__builtin_inff();
        h.seg = 0xFFFFFFFFu;
        h.doc = 0xFFFFFFFFu;
        oh[i] = h;
    }
    if (lane == 0) {
        out_nhits[q] = produced;
        out_found[q] = found;
    }
}

__global__ void __launch_bounds__(256) k_merge(const DevQuery* __restrict__ queries, uint32_t n_queries,
                                               const Hit* __restrict__ part_hits, const uint32_t* __restrict__ part_nhits,
                                               const uint64_t* __restrict__ part_found, Hit* __restrict__ out_hits,
                                               uint32_t* __restrict__ out_nhits, uint64_t* __restrict__ out_found,
                                               uint32_t K, uint32_t* __restrict__ heads /* one u32 per partial row */) {
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= n_queries) return;
    const DevQuery dq = queries[q];
    const uint32_t pb = dq.part_begin, pc = dq.part_count;
    if (merge_is_wide(pc, K)) return;   // k_merge_wide's
    uint64_t found = 0;
    for (uint32_t i = lane; i < pc; i += 64) found += part_found[pb + i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) found += __shfl_xor(found, d, 64);
    if (pc * K <= 64u) {
        // All entries of all rows fit the wave (the usual case: 1-6 rows at K = 10): one load per lane and one
        // 64-key bitonic sort on (score desc, (seg, doc) asc) instead of K rounds of dependent head loads.
        const uint32_t row = (uint32_t)lane / K, idx = (uint32_t)lane - row * K;
        uint32_t s = 0;            // order_bits(score); 0 == empty lane (sorts last)
        uint64_t id = ~0ull;
        if (row < pc && idx < part_nhits[pb + row]) {
            const Hit e = part_hits[(uint64_t)(pb + row) * K + idx];
            s = order_bits(e.score);
            id = ((uint64_t)e.seg << 32) | e.doc;
        }
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const uint32_t os = (uint32_t)__shfl_xor((int)s, j, 64);
                const uint64_t oid = __shfl_xor(id, j, 64);
                const bool other_better = os > s || (os == s && oid < id);
                const bool lower = (lane & j) == 0;            // this lane keeps the better of the pair in a best-first run
                const bool best_first = (lane & k) == 0;
                const bool take = (lower == best_first) ? other_better : !other_better && !(os == s && oid == id);
                if (take) { s = os; id = oid; }
            }
        const uint32_t produced = min((uint32_t)__popcll(__ballot(s != 0u)), K);
        if ((uint32_t)lane < K) {
            Hit h;
            if ((uint32_t)lane < produced) {
                h.score = unorder_bits(s);
                h.seg = (uint32_t)(id >> 32);
                h.doc = (uint32_t)id;
            } else {
                h.score = -__builtin_inff();
                h.seg = 0xFFFFFFFFu;
                h.doc = 0xFFFFFFFFu;
            }
            out_hits[(uint64_t)q * K + lane] = h;
        }
        // a query WITHOUT work items (no scored term: pc == 0) takes this path with any K: the rest of its row is padding too
        for (uint32_t i = 64u + (uint32_t)lane; i < K; i += 64u) {
            Hit h;
            h.score = -__builtin_inff();
            h.seg = 0xFFFFFFFFu;
            h.doc = 0xFFFFFFFFu;
            out_hits[(uint64_t)q * K + i] = h;
        }
        if (lane == 0) {
            out_nhits[q] = produced;
            out_found[q] = found;
        }
        return;
    }
    if (pc <= 64u && pc * K <= (uint32_t)kMergeStage) {
        // Up to 2048 entries in at most 64 rows (cfg3: 4-16 rows of K = 100): the scores' order bits are staged in LDS and
        // the K rounds run out of it — lane r keeps row r's position and head in registers, one DPP maximum picks the round's
        // score, the lowest lane that holds it wins (rows of a query ascend in (segment, doc range), and inside a row equal
        // scores ascend in docId: the canonical tie order), the winner re-reads its head from LDS.  The tournament below pays
        // two dependent GLOBAL round trips per round (row heads kept in memory): 100 rounds = 0.1 ms for a K = 100 batch.
        __shared__ uint32_t s_sc[4][kMergeStage];
        __shared__ uint16_t s_win[4][128];
        uint32_t* sc = s_sc[threadIdx.x >> 6];
        uint16_t* win = s_win[threadIdx.x >> 6];
        for (uint32_t e = lane; e < pc * K; e += 64) {
            const uint32_t r = e / K, i = e - r * K;
            sc[e] = (i < part_nhits[pb + r]) ? order_bits(part_hits[(uint64_t)(pb + r) * K + i].score) : 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t h = 0;
        uint32_t cur = ((uint32_t)lane < pc) ? sc[(uint32_t)lane * K] : 0u;   // 0 == nothing (left)
        uint32_t produced = 0;
        for (; produced < K; produced++) {
            const uint32_t ms = merge_wave_max(cur);
            if (ms == 0u) break;
            const uint32_t w = (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(cur == ms));
            if ((uint32_t)lane == w) {
                win[produced] = (uint16_t)((w << 7) | h);
                h++;
                cur = (h < K) ? sc[w * K + h] : 0u;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        Hit* oh = out_hits + (uint64_t)q * K;
        for (uint32_t i = lane; i < K; i += 64) {
            Hit hh;
            if (i < produced) {
                const uint32_t code = win[i];
                hh = part_hits[(uint64_t)(pb + (code >> 7)) * K + (code & 127u)];
            } else {
                hh.score = -__builtin_inff();
                hh.seg = 0xFFFFFFFFu;
                hh.doc = 0xFFFFFFFFu;
            }
            oh[i] = hh;
        }
        if (lane == 0) {
            out_nhits[q] = produced;
            out_found[q] = found;
        }
        return;
    }
    merge_rows_wave(q, pb, pc, part_hits, part_nhits, out_hits, out_nhits, out_found, found, K, heads, lane);
}

// Segment-sharded multi-GPU (SURVEY.md 8(e) alternative): every rank scored ALL queries over ITS segments; the
// per-rank rows were all-gathered rank-major ([rank][query][K]).  One wave per query, lane r holds rank r's row
// head; K rounds of wave argmax in the canonical order (score desc, GLOBAL seg asc, doc asc) rebuild the one
// global heap of src/api_engine.cpp:434-435,485-492; `found` is the sum over ranks (:495 counts per segment).
// seg_map[r * stride + local seg id] = the segment's id in the full manifest (nullptr: ids are already global).
__global__ void __launch_bounds__(256) k_merge_ranks(const Hit* __restrict__ hits, const uint32_t* __restrict__ nhits,
                                                     const uint64_t* __restrict__ found, uint32_t n_ranks, uint32_t n_queries, uint32_t K,
                                                     const uint32_t* __restrict__ seg_map, uint32_t seg_map_stride,
                                                     Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits, uint64_t* __restrict__ out_found) {
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= n_queries) return;
    const bool mine = (uint32_t)lane < n_ranks;
    const size_t row = (size_t)lane * n_queries + q;
    const uint32_t n = mine ? min(nhits[row], K) : 0u;
    uint64_t f = mine ? found[row] : 0ull;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) f += __shfl_xor(f, d, 64);
    const Hit* my = hits + row * K;
    uint32_t h = 0, produced = 0;
    Hit* oh = out_hits + (size_t)q * K;
    for (; produced < K; produced++) {
        uint32_t s = 0;
        uint64_t id = ~0ull;
        if (h < n) {
            const Hit e = my[h];
            const uint32_t gseg = seg_map ? seg_map[(size_t)lane * seg_map_stride + e.seg] : e.seg;
            s = order_bits(e.score);
            id = ((uint64_t)gseg << 32) | e.doc;
        }
        uint32_t ms = s;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) ms = max(ms, (uint32_t)__shfl_xor(ms, d, 64));
        if (__ballot(h < n) == 0ull) break;
        uint64_t mid = (h < n && s == ms) ? id : ~0ull;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const uint64_t o = __shfl_xor(mid, d, 64);
            mid = o < mid ? o : mid;
        }
        if (h < n && s == ms && id == mid) {   // (seg, doc) pairs are unique across ranks: one winner
            Hit w;
            w.score = unorder_bits(ms);
            w.seg = (uint32_t)(mid >> 32);
            w.doc = (uint32_t)mid;
            oh[produced] = w;
            h++;
        }
    }
    for (uint32_t i = produced + lane; i < K; i += 64) {
        Hit w;
        w.score = -__builtin_inff();
        w.seg = 0xFFFFFFFFu;
        w.doc = 0xFFFFFFFFu;
        oh[i] = w;
    }
    if (lane == 0) {
        out_nhits[q] = produced;
        out_found[q] = f;
    }
}

constexpr uint32_t kMergeCap = 2048;    // candidates held in LDS; more than that (mass ties) falls back to the tournament
constexpr uint32_t kMergeBins = 2048;   // 11 bits per histogram pass
constexpr uint32_t kMergeRegRows = 32;  // row heads cached per thread (a lone query has at most n_cus * 24 + a few rows)

// K-th largest over the histogram: s_sel[0] = bin holding it (0xFFFFFFFF: fewer than `target` entries), s_sel[1] =
// its rank inside that bin.  All 256 threads call; ends with a barrier.
__device__ __forceinline__ void merge_select_bin(uint32_t* s_hist, uint32_t* s_part, uint32_t* s_sel, uint32_t target, uint32_t tid) {
    uint32_t own = 0;
#pragma unroll
    for (uint32_t j = 0; j < kMergeBins / 256; j++) own += s_hist[tid * (kMergeBins / 256) + j];
    s_part[tid] = own;
    if (tid == 0) s_sel[0] = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t above = 0;
    for (uint32_t u = tid + 1; u < 256; u++) above += s_part[u];
    if (above < target && above + own >= target) {
        uint32_t acc = above;
        for (int j = (int)(kMergeBins / 256) - 1; j >= 0; j--) {
            const uint32_t c = s_hist[tid * (kMergeBins / 256) + j];
            if (acc + c >= target) { s_sel[0] = tid * (kMergeBins / 256) + (uint32_t)j; s_sel[1] = target - acc; break; }
            acc += c;
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_merge_wide(const DevQuery* __restrict__ queries, const uint32_t* __restrict__ wide_q,
                                                    const Hit* __restrict__ part_hits, const uint32_t* __restrict__ part_nhits,
                                                    const uint64_t* __restrict__ part_found, Hit* __restrict__ out_hits,
                                                    uint32_t* __restrict__ out_nhits, uint64_t* __restrict__ out_found,
                                                    uint32_t K, uint32_t* __restrict__ heads) {
    __shared__ uint32_t s_hist[kMergeBins];
    __shared__ uint32_t s_part[256];
    __shared__ uint32_t s_sel[2];
    __shared__ uint32_t s_ncand;
    __shared__ uint64_t s_fsum[4];
    __shared__ uint32_t s_cs[kMergeCap];
    __shared__ uint64_t s_cid[kMergeCap];
    const uint32_t tid = threadIdx.x;
    const uint32_t q = wide_q[blockIdx.x];
    const DevQuery dq = queries[q];
    const uint32_t pb = dq.part_begin, pc = dq.part_count;

    // row heads are read once: order_bits(head score) of this thread's first kMergeRegRows rows stay in registers
    // (0 == empty row); rows beyond that (a query cut into > 256 * kMergeRegRows ranges) are re-read.
    constexpr uint32_t R = kMergeRegRows;
    uint32_t hs[R];
    uint64_t found = 0;
#pragma unroll
    for (uint32_t r = 0; r < R; r++) {
        const uint32_t i = tid + r * 256;
        hs[r] = 0;
        if (i < pc) {
            found += part_found[pb + i];
            if (part_nhits[pb + i]) hs[r] = order_bits(part_hits[(uint64_t)(pb + i) * K].score);
        }
    }
    for (uint32_t i = tid + R * 256; i < pc; i += 256) found += part_found[pb + i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) found += __shfl_xor(found, d, 64);
    if ((tid & 63) == 0) s_fsum[tid >> 6] = found;
    if (tid == 0) s_ncand = 0;

    // theta: lower edge of the 22-bit bucket that holds the K-th largest row head
    for (uint32_t i = tid; i < kMergeBins; i += 256) s_hist[i] = 0;
    __syncthreads();
    found = s_fsum[0] + s_fsum[1] + s_fsum[2] + s_fsum[3];
#pragma unroll
    for (uint32_t r = 0; r < R; r++)
        if (hs[r]) atomicAdd(&s_hist[hs[r] >> 21], 1u);
    for (uint32_t i = tid + R * 256; i < pc; i += 256)
        if (part_nhits[pb + i]) atomicAdd(&s_hist[order_bits(part_hits[(uint64_t)(pb + i) * K].score) >> 21], 1u);
    __syncthreads();
    merge_select_bin(s_hist, s_part, s_sel, K, tid);
    const uint32_t b1 = s_sel[0], k2 = s_sel[1];
    uint32_t theta = 0;   // fewer than K rows hold anything: every entry is a candidate
    if (b1 != 0xFFFFFFFFu) {
        for (uint32_t i = tid; i < kMergeBins; i += 256) s_hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < R; r++)
            if (hs[r] && (hs[r] >> 21) == b1) atomicAdd(&s_hist[(hs[r] >> 10) & (kMergeBins - 1)], 1u);
        for (uint32_t i = tid + R * 256; i < pc; i += 256)
            if (part_nhits[pb + i]) {
                const uint32_t s = order_bits(part_hits[(uint64_t)(pb + i) * K].score);
                if ((s >> 21) == b1) atomicAdd(&s_hist[(s >> 10) & (kMergeBins - 1)], 1u);
            }
        __syncthreads();
        merge_select_bin(s_hist, s_part, s_sel, k2, tid);
        theta = (b1 << 21) | (s_sel[0] << 10);
    }

    // gather every row's prefix with score >= theta
    auto gather_row = [&](uint32_t i) {
        const uint32_t n = part_nhits[pb + i];
        const Hit* row = part_hits + (uint64_t)(pb + i) * K;
        for (uint32_t j = 0; j < n; j++) {
            const Hit e = row[j];
            const uint32_t s = order_bits(e.score);
            if (s < theta) break;
            const uint32_t slot = atomicAdd(&s_ncand, 1u);
            if (slot < kMergeCap) { s_cs[slot] = s; s_cid[slot] = ((uint64_t)e.seg << 32) | e.doc; }
        }
    };
#pragma unroll
    for (uint32_t r = 0; r < R; r++)
        if (hs[r] && hs[r] >= theta) gather_row(tid + r * 256);
    for (uint32_t i = tid + R * 256; i < pc; i += 256) gather_row(i);
    __syncthreads();
    const uint32_t C = s_ncand;
    if (C > kMergeCap) {   // uniform: one wave redoes the query the slow way
        if (tid < 64) merge_rows_wave(q, pb, pc, part_hits, part_nhits, out_hits, out_nhits, out_found, found, K, heads, (int)tid);
        return;
    }
    uint32_t P = 64;
    while (P < C) P <<= 1;
    for (uint32_t i = C + tid; i < P; i += 256) { s_cs[i] = 0; s_cid[i] = ~0ull; }
    __syncthreads();
    // bitonic sort, best first: (score bits desc, id asc)
    for (uint32_t k = 2; k <= P; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (P >> 1); t += 256) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const uint32_t sa = s_cs[lo], sb = s_cs[hi];
                const uint64_t ia = s_cid[lo], ib = s_cid[hi];
                const bool b_better = sb > sa || (sb == sa && ib < ia);
                const bool desc = (lo & k) == 0;   // this run wants best first
                if (b_better == desc) { s_cs[lo] = sb; s_cs[hi] = sa; s_cid[lo] = ib; s_cid[hi] = ia; }
            }
            __syncthreads();
        }
    const uint32_t produced = C < K ? C : K;
    Hit* oh = out_hits + (uint64_t)q * K;
    for (uint32_t i = tid; i < K; i += 256) {
        Hit h;
        if (i < produced) {
            h.score = unorder_bits(s_cs[i]);
            h.seg = (uint32_t)(s_cid[i] >> 32);
            h.doc = (uint32_t)s_cid[i];
        } else {
            h.score = -__builtin_inff();
            h.seg = 0xFFFFFFFFu;
            h.doc = 0xFFFFFFFFu;
        }
        oh[i] = h;
    }
    if (tid == 0) {
        out_nhits[q] = produced;
        out_found[q] = found;
    }
}

}  // namespace ns

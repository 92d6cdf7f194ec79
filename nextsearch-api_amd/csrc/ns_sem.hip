// SPDX-License-Identifier: MIT
// Semantic query expansion on the device: SemanticIndex::most_similar_to_vec,
// src/semantic_embedding.cpp:104-145 — for a query vector, the top-k rows of the L2-normalised embedding
// table by dot product, skipping banned rows and rows with sim < min_sim.  It produces the per-term fp32
// qweights the hot path already accepts (:186-189, :213-218).
//
// Bit-exactness: the reference's dot() is `s += a[i] * b[i]` in index order (:11-15), one rounding per
// multiply and one per add (the checked build has no FMA contraction; this library is compiled with
// -ffp-contract=off).  One thread owns one table row and walks the dimensions in order with kSemB query
// accumulators — the same sequence of fp32 operations per (query, row) — so the sims are the reference's
// bits and the top-k is exact (ties: the smaller row index, as the reference's `sim > heap.front()` keeps
// the earlier row).  The table is stored dimension-major ([dim][rows]) so that a wave's 64 rows are 64
// consecutive floats per dimension; the query values are wave-uniform (scalar loads).
//
// HBM-bound: one pass over the table (rows x dim x 4 B) per group of kSemB query vectors; the selection rides in the same
// kernel (k_sem_scan_topk) and a one-workgroup-per-query kernel picks the final top-k from the few keys it left.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ns {

constexpr int kSemB = 8;             // query vectors per pass over the table
constexpr int kSemMaxK = 64;

__device__ __forceinline__ uint32_t sem_order_bits(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// larger == better: higher sim, then smaller index
__device__ __forceinline__ uint64_t sem_key(float sim, uint32_t idx) { return ((uint64_t)sem_order_bits(sim) << 32) | (uint64_t)(~idx); }

__global__ void __launch_bounds__(256) k_sem_transpose(const float* __restrict__ in /* [rows][dim] */, float* __restrict__ out /* [dim][rows_pad] */,
                                                       uint32_t rows, uint32_t dim, uint32_t rows_pad) {
    __shared__ float tile[32][33];
    const uint32_t r0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint32_t r = r0 + j, d = d0 + tx;
        tile[j][tx] = (r < rows && d < dim) ? in[(size_t)r * dim + d] : 0.0f;
    }
    __syncthreads();
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint32_t d = d0 + j, r = r0 + tx;
        if (d < dim && r < rows_pad) out[(size_t)d * rows_pad + r] = tile[tx][j];
    }
}

// 64-bit wave maximum; every lane gets the result
__device__ __forceinline__ uint64_t sem_wave_max(uint64_t k) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint64_t o = __shfl_xor(k, d, 64);
        k = o > k ? o : k;
    }
    return k;
}

// The scan WITH the selection (round 3; round 2 wrote the kSemB x rows sims out and selected in two more kernels, 24 % of a
// group's time).  One thread owns one table row and walks the dimensions in order with kSemB query accumulators; then, per
// query, the wave's rows that pass (sim >= min_sim, :124; not banned, :119) hand their up-to-topk best keys to the query's
// candidate list — an atomic cursor per query; almost every wave has none and leaves after one ballot per query.
// cand[b][0 .. count[b]): keys in arrival order (they are unique: the final selection is by key, so the order is immaterial).
__global__ void __launch_bounds__(256) k_sem_scan_topk(const float* __restrict__ vt, uint32_t rows, uint32_t rows_pad, uint32_t dim,
                                                       const float* __restrict__ q /* [kSemB][dim] */, float min_sim,
                                                       const uint32_t* __restrict__ ban_off /* [kSemB + 1] */, const uint32_t* __restrict__ ban_rows,
                                                       uint32_t topk, uint32_t cap /* keys per query */, uint64_t* __restrict__ cand /* [kSemB][cap] */,
                                                       uint32_t* __restrict__ count /* [kSemB], zero at launch */) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    const bool live = r < rows;
    const uint32_t rr = live ? r : 0u;   // lanes past the table walk row 0 and offer nothing (the wave stays whole for the shuffles)
    float acc[kSemB];
#pragma unroll
    for (int b = 0; b < kSemB; b++) acc[b] = 0.0f;
    constexpr uint32_t U = 8;   // table loads in flight per lane (the accumulation below stays in index order)
    uint32_t i = 0;
    for (; i + U <= dim; i += U) {
        float v[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) v[u] = vt[(size_t)(i + u) * rows_pad + rr];
#pragma unroll
        for (uint32_t u = 0; u < U; u++)
#pragma unroll
            for (int b = 0; b < kSemB; b++) acc[b] = acc[b] + q[(size_t)b * dim + i + u] * v[u];   // :13 `s += a[i] * b[i]`
    }
    for (; i < dim; i++) {
        const float v = vt[(size_t)i * rows_pad + rr];
#pragma unroll
        for (int b = 0; b < kSemB; b++) acc[b] = acc[b] + q[(size_t)b * dim + i] * v;
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int b = 0; b < kSemB; b++) {
        bool ok = live && !(acc[b] < min_sim);                              // :124 (`sim < min_sim` skips; a NaN sim does not)
        if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;              // uniform: the usual case
        for (uint32_t j = ban_off[b]; j < ban_off[b + 1]; j++) ok = ok && ban_rows[j] != r;   // :119 (uniform bounds)
        uint64_t key = ok ? sem_key(acc[b], r) : 0ull;
        const uint32_t n = min((uint32_t)__popcll(__builtin_amdgcn_ballot_w64(ok)), topk);
        if (n == 0u) continue;
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(&count[b], n);
        pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
        for (uint32_t t = 0; t < n; t++) {                                  // the wave's n best, best first
            const uint64_t m = sem_wave_max(key);
            if (lane == 0 && pos + t < cap) cand[(size_t)b * cap + pos + t] = m;
            if (key == m) key = 0ull;                                       // keys are unique (distinct rows)
        }
    }
}

// workgroup argmax of a 64-bit key; every thread gets the result.  Two barriers.
__device__ __forceinline__ uint64_t sem_block_max(uint64_t k, uint64_t* s_red) {
    k = sem_wave_max(k);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = k;
    __syncthreads();
    uint64_t m = s_red[0];
#pragma unroll
    for (int j = 1; j < 4; j++) m = s_red[j] > m ? s_red[j] : m;
    return m;
}

// query: the best topk of the candidates its scan left
__global__ void __launch_bounds__(256) k_sem_final_topk(uint64_t* __restrict__ cand, uint32_t cap, uint32_t* __restrict__ count /* keys the scan left per query; reset here */,
                                                        uint32_t topk, uint32_t* __restrict__ rows_out, float* __restrict__ sims_out, uint32_t* __restrict__ count_out) {
    __shared__ uint64_t s_red[4];
    const uint32_t b = blockIdx.x;
    uint64_t* c = cand + (size_t)b * cap;
    const uint32_t n_cand = min(count[b], cap);
    uint32_t produced = 0;
    for (; produced < topk; produced++) {
        uint64_t best = 0;
        uint32_t at = 0;
        for (uint32_t j = threadIdx.x; j < n_cand; j += 256) {
            const uint64_t k = c[j];
            if (k > best) { best = k; at = j; }
        }
        const uint64_t m = sem_block_max(best, s_red);
        if (m == 0) break;
        if (best == m) {   // keys are unique (distinct rows): one owner
            c[at] = 0;
            const uint32_t ob = (uint32_t)(m >> 32);
            rows_out[(size_t)b * topk + produced] = ~(uint32_t)m;
            sims_out[(size_t)b * topk + produced] = __uint_as_float((ob & 0x80000000u) ? (ob ^ 0x80000000u) : ~ob);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { count_out[b] = produced; count[b] = 0; }   // for the next group (every thread read count[b] before the loop's first barrier)
}

}  // namespace ns

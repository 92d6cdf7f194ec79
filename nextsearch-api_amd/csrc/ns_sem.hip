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
// HBM-bound: one pass over the table (rows x dim x 4 B) per group of kSemB query vectors.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ns {

constexpr int kSemB = 8;             // query vectors per pass over the table
constexpr int kSemChunk = 8192;      // rows per selection workgroup
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

__global__ void __launch_bounds__(256) k_sem_sims(const float* __restrict__ vt, uint32_t rows, uint32_t rows_pad, uint32_t dim,
                                                  const float* __restrict__ q /* [kSemB][dim] */, float* __restrict__ sims /* [kSemB][rows_pad] */) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float acc[kSemB];
#pragma unroll
    for (int b = 0; b < kSemB; b++) acc[b] = 0.0f;
    constexpr uint32_t U = 8;   // table loads in flight per lane (the accumulation below stays in index order)
    uint32_t i = 0;
    for (; i + U <= dim; i += U) {
        float v[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) v[u] = vt[(size_t)(i + u) * rows_pad + r];
#pragma unroll
        for (uint32_t u = 0; u < U; u++)
#pragma unroll
            for (int b = 0; b < kSemB; b++) acc[b] = acc[b] + q[(size_t)b * dim + i + u] * v[u];   // :13 `s += a[i] * b[i]`
    }
    for (; i < dim; i++) {
        const float v = vt[(size_t)i * rows_pad + r];
#pragma unroll
        for (int b = 0; b < kSemB; b++) acc[b] = acc[b] + q[(size_t)b * dim + i] * v;
    }
#pragma unroll
    for (int b = 0; b < kSemB; b++) sims[(size_t)b * rows_pad + r] = acc[b];
}

// workgroup argmax of a 64-bit key; every thread gets the result.  Two barriers.
__device__ __forceinline__ uint64_t sem_block_max(uint64_t k, uint64_t* s_red) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint64_t o = __shfl_xor(k, d, 64);
        k = o > k ? o : k;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = k;
    __syncthreads();
    uint64_t m = s_red[0];
#pragma unroll
    for (int j = 1; j < 4; j++) m = s_red[j] > m ? s_red[j] : m;
    return m;
}

// (chunk, query): the chunk's topk rows with sim >= min_sim that are not banned (:119-124), best first.
__global__ void __launch_bounds__(256) k_sem_chunk_topk(const float* __restrict__ sims, uint32_t rows, uint32_t rows_pad, float min_sim,
                                                        const uint32_t* __restrict__ ban_off /* [kSemB + 1] */, const uint32_t* __restrict__ ban_rows,
                                                        uint32_t topk, uint32_t n_chunks, uint64_t* __restrict__ cand /* [kSemB][n_chunks][topk] keys, 0 = none */) {
    __shared__ float s[kSemChunk];
    __shared__ uint64_t s_red[4];
    const uint32_t chunk = blockIdx.x, b = blockIdx.y;
    const uint32_t base = chunk * (uint32_t)kSemChunk;
    const float ninf = -__builtin_inff();
    for (uint32_t j = threadIdx.x; j < (uint32_t)kSemChunk; j += 256) {
        const uint32_t r = base + j;
        float v = ninf;
        if (r < rows) {
            v = sims[(size_t)b * rows_pad + r];
            if (v < min_sim) v = ninf;                                   // :124
        }
        s[j] = v;
    }
    __syncthreads();
    for (uint32_t j = ban_off[b] + threadIdx.x; j < ban_off[b + 1]; j += 256) {   // :119
        const uint32_t r = ban_rows[j];
        if (r >= base && r < base + (uint32_t)kSemChunk) s[r - base] = ninf;
    }
    __syncthreads();
    uint64_t* out = cand + ((size_t)b * n_chunks + chunk) * topk;
    for (uint32_t round = 0; round < topk; round++) {
        uint64_t best = 0;
        for (uint32_t j = threadIdx.x; j < (uint32_t)kSemChunk; j += 256) {
            const float v = s[j];
            if (v != ninf) {
                const uint64_t k = sem_key(v, base + j);
                best = k > best ? k : best;
            }
        }
        best = sem_block_max(best, s_red);
        if (threadIdx.x == 0) out[round] = best;
        if (best == 0) {   // uniform: nothing left
            for (uint32_t t = round + 1 + threadIdx.x; t < topk; t += 256) out[t] = 0;
            break;
        }
        if (threadIdx.x == 0) s[(~(uint32_t)best) - base] = ninf;
        __syncthreads();
    }
}

// query: the best topk of its chunks' candidates
__global__ void __launch_bounds__(256) k_sem_final_topk(uint64_t* __restrict__ cand, uint32_t n_cand /* n_chunks * topk */, uint32_t topk,
                                                        uint32_t* __restrict__ rows_out, float* __restrict__ sims_out, uint32_t* __restrict__ count_out) {
    __shared__ uint64_t s_red[4];
    const uint32_t b = blockIdx.x;
    uint64_t* c = cand + (size_t)b * n_cand;
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
    if (threadIdx.x == 0) count_out[b] = produced;
}

}  // namespace ns

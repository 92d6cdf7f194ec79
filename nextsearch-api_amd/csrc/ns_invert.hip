// SPDX-License-Identifier: MIT
// Index inversion on the device: forward.bin -> posting lists in termId order, each sorted
// by docId — the step immediately before the hot path.  The reference (src/lexicon.cpp:52-128) appends every
// (termId, tf) of every document to a per-term std::vector and std::sorts each vector by docId.
//
// Here: forward.bin lists documents in docId order, so a STABLE sort of the (termId, tf) pairs by termId
// alone yields every list already sorted by docId.  That is an LSD radix sort with 8-bit digits over
// ceil(log2(n_terms + 1) / 8) passes (2 for a 65 536-term vocabulary, 3 up to 16 M terms):
//
//   k_iv_expand   pair i -> key = termId (n_terms for the ids the reference drops, :70), value = {docId, tf};
//                 docId by binary search in the prefix sums of the per-document counts
//   per pass:     k_iv_hist (LDS histogram per 4096-pair tile, written digit-major) -> exclusive scan of the
//                 256 x tiles counters -> k_iv_scatter (stable ranks: wave-level match masks from 8 ballots,
//                 per-wave digit counters in LDS, waves of a tile ordered by a 256-thread prefix)
//   k_iv_run_starts / k_iv_run_lengths   df[t] = length of term t's run in the sorted keys
//
// Integer work only; the result is defined bit for bit (the one freedom the reference leaves — the order of
// equal docIds inside a list, std::sort being unstable — is resolved as input order).
// Algorithmic bytes per pair: 8 B read (termId, tf) + 8 B written (docId, tf) = 16 B; the radix passes move
// 12 B in + 12 B out per pair per pass plus 4 B for the histogram read.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ns {

constexpr int kIvItems = 16;                 // pairs per thread per tile
constexpr int kIvTile = 256 * kIvItems;      // pairs per workgroup

__global__ void __launch_bounds__(256) k_iv_expand(const uint2* __restrict__ pairs, const uint64_t* __restrict__ doc_prefix,
                                                   uint32_t n_docs, uint32_t n_pairs, uint32_t n_terms,
                                                   uint32_t* __restrict__ keys, uint2* __restrict__ vals) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += gridDim.x * 256) {
        // the document that holds pair i: the last d with doc_prefix[d] <= i (documents may be empty)
        uint32_t lo = 0, hi = n_docs;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (doc_prefix[mid + 1] <= (uint64_t)i) lo = mid + 1; else hi = mid;
        }
        const uint2 p = pairs[i];   // {termId, tf}
        const bool keep = p.x < n_terms;
        keys[i] = keep ? p.x : n_terms;
        vals[i] = make_uint2(lo, p.y);
    }
}

// df from the SORTED keys (a histogram by atomics serialises on the frequent terms: the most frequent one
// occurs in almost every document): a run's first pair records where it starts, its last pair the length.
__global__ void __launch_bounds__(256) k_iv_run_starts(const uint32_t* __restrict__ keys, uint32_t n, uint32_t n_terms, uint32_t* __restrict__ first) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    if (k < n_terms && (i == 0 || keys[i - 1] != k)) first[k] = i;
}
__global__ void __launch_bounds__(256) k_iv_run_lengths(const uint32_t* __restrict__ keys, uint32_t n, uint32_t n_terms,
                                                        const uint32_t* __restrict__ first, uint32_t* __restrict__ df) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    if (k < n_terms && (i + 1 == n || keys[i + 1] != k)) df[k] = i + 1 - first[k];
}

__global__ void __launch_bounds__(256) k_iv_hist(const uint32_t* __restrict__ keys, uint32_t n, uint32_t shift,
                                                 uint32_t* __restrict__ tile_hist /* [256][n_tiles] */, uint32_t n_tiles) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)kIvTile;
#pragma unroll
    for (int s = 0; s < kIvItems; s++) {
        const uint32_t idx = base + (uint32_t)s * 256 + threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    tile_hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// ---- exclusive scan of a flat u32 array (three small kernels; 1024 elements per workgroup) ----
__global__ void __launch_bounds__(256) k_iv_scan_sums(const uint32_t* __restrict__ a, uint32_t m, uint32_t* __restrict__ sums) {
    __shared__ uint32_t red[4];
    const uint32_t base = blockIdx.x * 1024u;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = base + threadIdx.x * 4 + j;
        if (i < m) s += a[i];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(1024) k_iv_scan_top(uint32_t* __restrict__ sums, uint32_t n) {   // one workgroup, in place, exclusive
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? sums[i] : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {   // Hillis-Steele, inclusive
            const uint32_t t = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        const uint32_t c = carry;
        if (i < n) sums[i] = c + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + part[1023];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_iv_scan_apply(uint32_t* __restrict__ a, uint32_t m, const uint32_t* __restrict__ sums) {
    __shared__ uint32_t wsum[4];
    const uint32_t base = blockIdx.x * 1024u;
    uint32_t v[4], t = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = base + threadIdx.x * 4 + j;
        v[j] = i < m ? a[i] : 0u;
        t += v[j];
    }
    // exclusive prefix of t over the workgroup: inclusive wave scan, then the waves' totals
    uint32_t inc = t;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t off = sums[blockIdx.x] + inc - t;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) off += wsum[w];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = base + threadIdx.x * 4 + j;
        if (i < m) a[i] = off;
        off += v[j];
    }
}

// Stable scatter of one tile.  Wave w owns the tile's pairs [w*1024, (w+1)*1024) in 16 steps of 64; inside a
// step the pairs with the same digit find each other with 8 ballots; ranks continue from the wave's running
// per-digit counter in LDS (LDS operations of one wave execute in order).
__global__ void __launch_bounds__(256) k_iv_scatter(const uint32_t* __restrict__ keys_in, const uint2* __restrict__ vals_in,
                                                    uint32_t* __restrict__ keys_out, uint2* __restrict__ vals_out, uint32_t n,
                                                    uint32_t shift, const uint32_t* __restrict__ tile_base /* scanned [256][n_tiles] */,
                                                    uint32_t n_tiles) {
    __shared__ uint32_t wcnt[4][256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 4; j++) wcnt[j][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)kIvTile + (uint32_t)w * (64u * kIvItems);
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t key[kIvItems], rank[kIvItems];
#pragma unroll
    for (int s = 0; s < kIvItems; s++) {
        const uint32_t idx = base + (uint32_t)s * 64 + (uint32_t)lane;
        const bool valid = idx < n;
        key[s] = valid ? keys_in[idx] : 0xFFFFFFFFu;
        const uint32_t d = (key[s] >> shift) & 255u;
        uint64_t mask = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __builtin_amdgcn_ballot_w64(bit);
            mask &= bit ? bal : ~bal;
        }
        const uint32_t prev = wcnt[w][d];
        rank[s] = prev + (uint32_t)__popcll(mask & lt);
        if (valid && (mask & lt) == 0ull) wcnt[w][d] = prev + (uint32_t)__popcll(mask);   // the group's first lane
    }
    __syncthreads();
    {   // thread d: where each wave's pairs of digit d start in the output
        const uint32_t d = threadIdx.x;
        uint32_t run = tile_base[(size_t)d * n_tiles + blockIdx.x];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t c = wcnt[j][d];
            wcnt[j][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kIvItems; s++) {
        const uint32_t idx = base + (uint32_t)s * 64 + (uint32_t)lane;
        if (idx < n) {
            const uint32_t pos = wcnt[w][(key[s] >> shift) & 255u] + rank[s];
            keys_out[pos] = key[s];
            vals_out[pos] = vals_in[idx];
        }
    }
}

}  // namespace ns

// SPDX-License-Identifier: MIT
// Index inversion on the device: forward.bin -> posting lists in termId order, each sorted
// by docId — the step immediately before the hot path.  The reference (src/lexicon.cpp:52-128) appends every
// (termId, tf) of every document to a per-term std::vector and std::sorts each vector by docId.
//
// Here: forward.bin lists documents in docId order, so a STABLE sort of the (termId, tf) pairs by termId
// alone yields every list already sorted by docId.  That is an LSD radix sort over the bits of n_terms - 1 in
// ceil(bits / 11) passes of 8 .. 11 bits (2 passes for every vocabulary up to 4 M terms):
//
//   per pass:     k_iv_hist_w (LDS histogram per 4096-item tile, written digit-major; the first pass reads the pairs
//                 themselves) -> exclusive scan of the BINS x tiles counters -> k_iv_pass (stable ranks: wave-level match
//                 masks from ballots, per-wave digit counters in LDS, the tile put in digit order in LDS before it leaves)
//   first pass:   makes the docIds on the way (prefix sums of the per-document counts, boundaries marked per tile in LDS)
//                 and lets the pairs the reference drops (:69-70) fall out: they are neither ranked nor written
//   k_iv_runs     first / last position of every term's run in the sorted keys (df = last - first + 1)
//
// Integer work only; the result is defined bit for bit (the one freedom the reference leaves — the order of
// equal docIds inside a list, std::sort being unstable — is resolved as input order).
// Algorithmic bytes per pair: 8 B read (termId, tf) + 8 B written (docId, tf) = 16 B.  Moved, two passes: 8 (histogram
// over the pairs) + 8 + 12 (first pass) + 4 (histogram over the keys) + 12 + 12 (second pass) + 4 (runs) + the digit
// counters = ~62 B; round 2's expand + two or three 8-bit passes moved 80 or 108 B.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ns {

constexpr int kIvItems = 16;                 // pairs per thread per tile
constexpr int kIvTile = 256 * kIvItems;      // pairs per workgroup

// df from the SORTED keys (a histogram by atomics serialises on the frequent terms: the most frequent one
// occurs in almost every document): a run's first pair records where it starts, its last pair where it ends;
// the host subtracts.
__global__ void __launch_bounds__(256) k_iv_runs(const uint32_t* __restrict__ keys, uint32_t n_arg, uint32_t n_terms,
                                                 uint32_t* __restrict__ first, uint32_t* __restrict__ last, const uint32_t* __restrict__ n_dev = nullptr) {
    const uint32_t n = n_dev ? *n_dev : n_arg;
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 4u;   // four consecutive keys per thread: one 16-byte load + the two neighbours
    if (i0 >= n) return;
    uint32_t k[6];
    k[0] = i0 ? keys[i0 - 1] : 0xFFFFFFFFu;
    if (i0 + 4 <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(keys + i0);
        k[1] = v.x; k[2] = v.y; k[3] = v.z; k[4] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) k[1 + j] = i0 + (uint32_t)j < n ? keys[i0 + j] : 0xFFFFFFFFu;
    }
    k[5] = i0 + 4 < n ? keys[i0 + 4] : 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = i0 + (uint32_t)j;
        if (i >= n || k[1 + j] >= n_terms) continue;
        if (i == 0 || k[j] != k[1 + j]) first[k[1 + j]] = i;
        if (i + 1 == n || k[2 + j] != k[1 + j]) last[k[1 + j]] = i;
    }
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
__global__ void __launch_bounds__(1024) k_iv_scan_top(uint32_t* __restrict__ sums, uint32_t n, uint32_t* __restrict__ total_out = nullptr) {   // one workgroup, in place, exclusive; total_out (optional): the sum of all
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
    if (total_out && threadIdx.x == 0) *total_out = carry;
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


// ---- the passes (round 3: no separate expand pass, digits as wide as the keys need, dropped pairs out in the first pass;
//      the pass count follows from n_terms alone — round 2 needed a device -> host sync on the number of dropped pairs
//      to choose between 2 and 3 passes.  Counting df with one atomicAdd per run of equal keys of a tile in the last pass
//      was built and measured: 10 M atomics, 2.8 ms — the sorted keys and k_iv_runs stay) ----

template <int BITS>
__global__ void __launch_bounds__(256) k_iv_hist_w(const uint32_t* __restrict__ keys, const uint2* __restrict__ pairs, uint32_t n_arg,
                                                    const uint32_t* __restrict__ n_dev /* number of items, when only the device knows it */,
                                                    uint32_t n_terms, uint32_t shift, uint32_t* __restrict__ tile_hist /* [BINS][n_tiles] */,
                                                    uint32_t n_tiles) {
    constexpr int BINS = 1 << BITS;
    __shared__ uint32_t h[BINS];
    const uint32_t n = n_dev ? *n_dev : n_arg;
    for (int d = threadIdx.x; d < BINS; d += 256) h[d] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)kIvTile + threadIdx.x * (uint32_t)kIvItems;
    // the thread's 16 consecutive keys (a pair's key is its termId; a termId the reference drops, src/lexicon.cpp:69-70, is
    // not counted: such a pair leaves the sort in the first pass); 16-byte loads (base is a multiple of 16 items)
    uint32_t k[kIvItems];
    if (base + kIvItems <= n) {
        if (pairs) {
            const uint4* p = reinterpret_cast<const uint4*>(pairs + base);
#pragma unroll
            for (int j = 0; j < kIvItems / 2; j++) { const uint4 v = p[j]; k[2 * j] = v.x; k[2 * j + 1] = v.z; }
        } else {
            const uint4* p = reinterpret_cast<const uint4*>(keys + base);
#pragma unroll
            for (int j = 0; j < kIvItems / 4; j++) { const uint4 v = p[j]; k[4 * j] = v.x; k[4 * j + 1] = v.y; k[4 * j + 2] = v.z; k[4 * j + 3] = v.w; }
        }
    } else {
#pragma unroll
        for (int j = 0; j < kIvItems; j++) {
            const uint32_t i = base + (uint32_t)j;
            k[j] = i < n ? (pairs ? pairs[i].x : keys[i]) : 0xFFFFFFFFu;
        }
    }
    uint32_t d = 0xFFFFFFFFu, run = 0;   // one LDS atomic per run of equal digits among the thread's keys
#pragma unroll
    for (int j = 0; j < kIvItems; j++) {
        const bool counted = pairs ? k[j] < n_terms : (base + (uint32_t)j < n);
        if (!counted) continue;
        const uint32_t dj = (k[j] >> shift) & (uint32_t)(BINS - 1);
        if (dj == d) { run++; } else { if (run) atomicAdd(&h[d], run); d = dj; run = 1; }
    }
    if (run) atomicAdd(&h[d], run);
    __syncthreads();
    for (int dd = threadIdx.x; dd < BINS; dd += 256) tile_hist[(size_t)dd * n_tiles + blockIdx.x] = h[dd];
}

// One pass over one tile: stable partition by the digit (key >> shift) & (BINS - 1).
//   FIRST: the input is forward.bin's pairs (key = termId; a pair whose termId the reference drops, src/lexicon.cpp:69-70,
//          takes no part: it is neither ranked nor written, so the keys need no extra bit for it; docIds are made here)
// n_in = items of the input (pairs, or what the previous pass wrote); df comes from the sorted keys afterwards (k_iv_runs)
template <int BITS, bool FIRST, bool LAST>
__global__ void __launch_bounds__(256) k_iv_pass(const uint2* __restrict__ pairs, const uint64_t* __restrict__ doc_prefix,
                                                 const uint2* __restrict__ tile_docs /* FIRST: per tile {doc of its first pair, doc of its last pair} */,
                                                 uint32_t n_terms, const uint32_t* __restrict__ keys_in, const uint2* __restrict__ vals_in,
                                                 uint32_t* __restrict__ keys_out, uint2* __restrict__ vals_out, uint32_t n_arg,
                                                 const uint32_t* __restrict__ n_dev, uint32_t shift,
                                                 const uint32_t* __restrict__ tile_base /* scanned [BINS][n_tiles] */, uint32_t n_tiles) {
    constexpr int BINS = 1 << BITS;
    const uint32_t n = n_dev ? *n_dev : n_arg;
    constexpr int DPT = BINS / 256;              // digits per thread in the digit-start scan
    __shared__ uint32_t wcnt[4][BINS];
    __shared__ uint32_t gdelta[BINS];             // where digit d's run of this tile starts in the output, minus its start in the tile
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t s_key[kIvTile];           // FIRST: the tile's docIds until the keys are staged here
    __shared__ uint2 s_val[kIvTile];
    __shared__ uint32_t s_wmax[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int d = threadIdx.x; d < 4 * BINS; d += 256) (&wcnt[0][0])[d] = 0;
    const uint32_t tile0 = blockIdx.x * (uint32_t)kIvTile;
    if (tile0 >= n) return;   // (a later pass is launched for the first pass's item count: the dropped pairs' tiles are empty)
    if (FIRST) {
        // docIds: the documents that START inside the tile mark their first pair; a running maximum gives every pair its doc
        uint32_t* s_doc = s_key;
#pragma unroll
        for (int s = 0; s < kIvItems; s++) s_doc[s * 256 + threadIdx.x] = 0;
        // (the documents of the tile's first and last pair come from the host, which holds the prefix sums: two binary
        //  searches per tile on the device were two threads' chains of ~20 dependent loads in front of everything else)
        const uint2 td = tile_docs[blockIdx.x];
        const uint32_t d0 = td.x, d1 = td.y;
        __syncthreads();   // the zeroes above are in place
        if (threadIdx.x == 0) s_doc[0] = d0;
        for (uint32_t d = d0 + 1 + threadIdx.x; d <= d1; d += 256) atomicMax(&s_doc[(uint32_t)(doc_prefix[d] - tile0)], d);
        __syncthreads();
        uint32_t m[kIvItems], run = 0;
#pragma unroll
        for (int j = 0; j < kIvItems; j++) { run = max(run, s_doc[threadIdx.x * kIvItems + j]); m[j] = run; }
        uint32_t inc = run;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t o = __shfl_up(inc, dd, 64);
            if (lane >= dd) inc = max(inc, o);
        }
        if (lane == 63) s_wmax[w] = inc;
        const uint32_t before_lane = __shfl_up(inc, 1, 64);
        __syncthreads();
        uint32_t before = lane ? before_lane : 0u;
        for (int j = 0; j < w; j++) before = max(before, s_wmax[j]);
#pragma unroll
        for (int j = 0; j < kIvItems; j++) s_doc[threadIdx.x * kIvItems + j] = max(m[j], before);
    }
    __syncthreads();
    const uint32_t base = tile0 + (uint32_t)w * (64u * kIvItems);
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t key[kIvItems], rank[kIvItems];
    uint2 val[kIvItems];
#pragma unroll
    for (int s = 0; s < kIvItems; s++) {
        const uint32_t idx = base + (uint32_t)s * 64 + (uint32_t)lane;
        const bool valid = idx < n;
        if (FIRST) {
            const uint2 p = valid ? pairs[idx] : make_uint2(0xFFFFFFFFu, 0u);
            key[s] = (valid && p.x < n_terms) ? p.x : 0xFFFFFFFFu;   // ~0: not part of the sort (past the end, or a dropped termId)
            val[s] = make_uint2(valid ? s_key[idx - tile0] : 0u, p.y);
        } else {
            key[s] = valid ? keys_in[idx] : 0xFFFFFFFFu;
            val[s] = valid ? vals_in[idx] : make_uint2(0u, 0u);
        }
    }
#pragma unroll
    for (int s = 0; s < kIvItems; s++) {
        const bool valid = key[s] != 0xFFFFFFFFu;
        const uint32_t d = (key[s] >> shift) & (uint32_t)(BINS - 1);
        uint64_t mask = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __builtin_amdgcn_ballot_w64(bit);
            mask &= bit ? bal : ~bal;
        }
        const uint32_t prev = wcnt[w][d];
        rank[s] = prev + (uint32_t)__popcll(mask & lt);
        if (valid && (mask & lt) == 0ull) wcnt[w][d] = prev + (uint32_t)__popcll(mask);   // the group's first lane
    }
    __syncthreads();   // (FIRST: every docId has been read out of s_key by now)
    {   // thread t: digits [t * DPT, (t + 1) * DPT): their starts in the tile (exclusive scan of the digit totals), the waves' starts inside
        uint32_t c[DPT][4], tot[DPT], tsum = 0;
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            tot[q] = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { c[q][j] = wcnt[j][threadIdx.x * DPT + q]; tot[q] += c[q][j]; }
            tsum += tot[q];
        }
        uint32_t inc = tsum;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t o = __shfl_up(inc, dd, 64);
            if (lane >= dd) inc += o;
        }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        uint32_t ex = inc - tsum;
        for (int j = 0; j < w; j++) ex += wtot[j];
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            const uint32_t d = threadIdx.x * DPT + q;
            gdelta[d] = tile_base[(size_t)d * n_tiles + blockIdx.x] - ex;
            uint32_t run = ex;
#pragma unroll
            for (int j = 0; j < 4; j++) { wcnt[j][d] = run; run += c[q][j]; }
            ex += tot[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kIvItems; s++) {
        if (key[s] != 0xFFFFFFFFu) {
            const uint32_t lp = wcnt[w][(key[s] >> shift) & (uint32_t)(BINS - 1)] + rank[s];
            s_key[lp] = key[s];
            s_val[lp] = val[s];
        }
    }
    __syncthreads();
    const uint32_t staged = wtot[0] + wtot[1] + wtot[2] + wtot[3];   // the tile's items that take part (FIRST: without the dropped pairs)
#pragma unroll
    for (int s = 0; s < kIvItems; s++) {
        const uint32_t i = (uint32_t)s * 256 + threadIdx.x;
        if (i < staged) {
            const uint32_t k = s_key[i];
            const uint32_t pos = gdelta[(k >> shift) & (uint32_t)(BINS - 1)] + i;
            keys_out[pos] = k;
            vals_out[pos] = s_val[i];
        }
    }
}

}  // namespace ns

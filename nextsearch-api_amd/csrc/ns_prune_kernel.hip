// pscore_body — single-term items with BLOCK-MAX pruning ("block-max scores"; ns_ctx_use_pruning).
//
// The reference walks every list from its first posting to its last (src/api_engine.cpp:470-481) because `found` (:495)
// is the size of the union of the lists' docs.  For a term group of ONE list that size is known without reading anything:
// every posting of the list is a doc of its own, so the item's `found` is the number of the list's postings inside its doc
// range.  What is left to compute is the item's top-K (:485-492), and for that a block of 256 postings only has to be read
// when its best possible score can still enter the top-K:
//   * ns_segment_build_blockmax stores, per 256 postings of a registered list, the largest term score
//     s = (idf * (tf * 2.2f)) / (tf + norm) of the block — the expression of :477-479 with the list's idf, evaluated with the
//     same fp32 operations (k_blockmax below);
//   * a doc's score in a single-term group is 0.0f + w * s (:480); for w > 0 the fp32 product is monotone in s, so
//     w * blockmax is an upper bound that some posting of the block attains exactly;
//   * blocks are visited in docId order and ties go to the smaller docId, so a block whose bound is <= theta (the score of
//     the item's current K-th best, all of whose members have smaller docIds) cannot contribute: it is skipped unread.
// A wave looks at 64 block maxima at a time (one load), ballots the live ones, and scores only those — with the code of
// the driver stream: scalar-base loads, the exact division, offers into the candidate buffer.  The result is bit-identical
// to the exhaustive path (same hits, same order, same found); tests run every golden and the full-size digests both ways.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ns_internal.h"
#include "ns_wave_kernel.hip"

namespace ns {

// One wave per block of kBmxBlock postings of a registered list: the block's largest term score (see above).
// starts / counts / idfs / entry: per list; blockIdx.y = list, blockIdx.x strides over the list's blocks.
__global__ void __launch_bounds__(64) k_blockmax(const uint2* __restrict__ postings, const float* __restrict__ pnorm,
                                                 float* __restrict__ out, const uint32_t* __restrict__ starts,
                                                 const uint32_t* __restrict__ counts, const float* __restrict__ idfs,
                                                 const uint32_t* __restrict__ entry) {
    const uint32_t l = blockIdx.y;
    const uint32_t first = starts[l], count = counts[l];
    const float idf = idfs[l];
    const uint32_t n_blocks = (count + kBmxBlock - 1) / kBmxBlock;
    const uint32_t lane = threadIdx.x;
    for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        uint32_t best = 0u;   // order_bits: larger float <=> larger key; 0 sorts below every real score
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t i = b * kBmxBlock + (uint32_t)c * 64u + lane;
            if (i < count) {
                const uint2 pv = postings[(uint64_t)first + i];
                const float tf = (float)pv.y;
                const float den = tf + pnorm[(uint64_t)first + i];
                const float num = idf * (tf * (1.2f + 1.0f));
                best = max(best, order_bits(num / den));   // the compiler's IEEE division: the bits the scoring kernels produce
            }
        }
        best = wave_max_dpp(best);
        if (lane == 0) out[entry[l] + b] = unorder_bits(best);
    }
}

template <bool AND, int CB>
__device__ __forceinline__ void pscore_body(const DevWItem& it, const DevTerm* __restrict__ terms, const DevSeg* __restrict__ segs,
                                            uint64_t* cand, Hit* __restrict__ out_hits, uint32_t* __restrict__ out_nhits,
                                            uint64_t* __restrict__ out_found, uint32_t K, const int lane) {
    constexpr int DE = 4;
    const DevSeg seg = segs[it.seg];
    const bool fast_div = (__builtin_amdgcn_readfirstlane((int)it.whole) & 8) != 0;
    const gp_u2 postings = (gp_u2)seg.postings;
    const gp_f32 pnorm = (gp_f32)seg.pnorm;
    // the group's one term (the host sets bit 7 only for term_count == 1, idf > 0, weight > 0, a registered list)
    uint32_t cur = 0, end = 0, base = 0, idf_bits = 0, wq_bits = 0, bmx0 = 0;
    if (lane == 0) {
        const DevTerm tm = terms[it.term_begin];
        base = (uint32_t)tm.list_off;
        idf_bits = __float_as_uint(tm.idf);
        wq_bits = __float_as_uint(tm.weight);
        bmx0 = tm.bmx - 1u;
        end = tm.count;
        if (!(it.whole & 1u)) {
            if ((it.whole & 64u) && tm.skip != 0u) {
                const gp_u32 sk = (gp_u32)seg.skips + (tm.skip - 1u);
                cur = sk[it.doc_lo / kSkipDocs] - base;
                end = sk[(it.doc_hi + (kSkipDocs - 1u)) / kSkipDocs] - base;
            } else {
                list_range(seg.postings + tm.list_off, tm.count, it.doc_lo, it.doc_hi, seg.n_docs, cur, end);
            }
            if (end < cur) end = cur;
        }
    }
    const uint32_t l_base = rdlane(base, 0);
    const uint32_t r_cur = rdlane(cur, 0), r_end = rdlane(end, 0);   // list-relative posting range of this item
    const float d_idf = __uint_as_float(rdlane(idf_bits, 0));
    const float d_wq = __uint_as_float(rdlane(wq_bits, 0));
    const gp_f32 bmx = (gp_f32)seg.blockmax + rdlane(bmx0, 0);

    float theta = -__builtin_inff();
    uint32_t ncand = 0, nsorted = 0;
    uint32_t n_read = 0;   // blocks actually read (diagnostic builds)
    (void)n_read;
    wave_sync();

    if (r_end > r_cur) {
        const uint32_t blk_first = r_cur / kBmxBlock, blk_last = (r_end - 1u) / kBmxBlock;
        for (uint32_t bb = blk_first; bb <= blk_last; bb += 64u) {
            const uint32_t myb = bb + (uint32_t)lane;
            float bound = -__builtin_inff();
            if (myb <= blk_last) bound = d_wq * bmx[myb];   // the score the block's best posting has: the fp32 product is monotone for w > 0
            uint64_t live = wballot(bound > theta);
            while (live != 0ull) {
                const uint32_t l = (uint32_t)__builtin_ctzll(live);
                live &= live - 1ull;
                if (!(__uint_as_float(rdlane(__float_as_uint(bound), l)) > theta)) continue;   // theta has risen since the ballot
                const uint32_t b = bb + l;
                const uint32_t p0 = max(r_cur, b * kBmxBlock), p1 = min(r_end, (b + 1u) * kBmxBlock);
                const uint32_t n = p1 - p0;   // 1 .. 256
                const gp_u2 sp = postings + ((size_t)l_base + p0);
                const gp_f32 np = pnorm + ((size_t)l_base + p0);
                nat_u2 ps[DE];
                float nr[DE], dx[DE];
#pragma unroll
                for (int j = 0; j < DE; j++) {   // scalar base + fixed lane offset; the buffers are padded past the last list
                    ps[j] = sp[j * 64 + lane];
                    nr[j] = np[j * 64 + lane];
                }
                {   // src/api_engine.cpp:477-480, operation for operation
                    float num[DE], den[DE];
#pragma unroll
                    for (int j = 0; j < DE; j++) {
                        const float tf = (float)ps[j].y;
                        den[j] = tf + nr[j];
                        num[j] = d_idf * (tf * (1.2f + 1.0f));
                    }
                    ns_div_n<DE>(dx, num, den, fast_div);
#pragma unroll
                    for (int j = 0; j < DE; j++) dx[j] = d_wq * dx[j];   // 0.0f + w*s == w*s: w*s >= +0 here (idf > 0, w > 0)
                }
#pragma unroll
                for (int j = 0; j < DE; j++) {
                    const uint32_t left = (n > (uint32_t)(j * 64)) ? (n - (uint32_t)(j * 64)) : 0u;   // scalar
                    if (left == 0u) continue;
                    const uint64_t nmask = (left >= 64u) ? ~0ull : ((1ull << left) - 1ull);
                    // docIds ascend from block to block and inside a block: whoever is in the buffer has a smaller docId than
                    // the posting offered now, so a tie with theta loses — `>` is exact (no `>=` mode in this body)
                    uint64_t m = wballot(dx[j] > theta) & nmask;
                    if (m == 0ull) continue;
                    uint32_t c = (uint32_t)__popcll(m);
                    if (ncand + c > (uint32_t)CB) {
                        ncand = wave_shrink_cb<CB>(cand, ncand, nsorted, theta, K, lane);
                        m = wballot(dx[j] > theta) & nmask;
                        c = (uint32_t)__popcll(m);
                    }
                    if (__builtin_amdgcn_inverse_ballot_w64(m)) cand[ncand + lanes_below(m)] = make_key(dx[j], ps[j].x);
                    ncand += c;
                }
                // theta is what prunes: raise it as soon as the buffer holds K candidates that the last shrink has not seen
                if (ncand >= K && ncand - nsorted >= max(16u, K >> 2)) ncand = wave_shrink_cb<CB>(cand, ncand, nsorted, theta, K, lane);
            }
        }
    }

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
        out_found[it.out_slot] = (uint64_t)(r_end - r_cur);   // one list: every posting of the range is a doc of its own (:495)
    }
}

}  // namespace ns

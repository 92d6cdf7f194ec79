// Deterministic synthetic "CORD-19-shaped" index generator.
//
// Writes a complete index directory in the reference's on-disk format (see index_format.hpp for the
// file:line authority of every file) so that the reference engine, the oracle and this engine all
// load the same bytes.  Integer-only sampling (own SplitMix64/xoshiro256**, fixed-point tables; no
// <random> distributions, no libm) so that every machine produces identical files.
//
//   doc_len   : e ~ U{9..13}, m ~ U[0,2^e): len = 2^e + m; with p = 1/16 replaced by U[20,511]
//   vocabulary: V terms, rank r = 1..V, termId = r-1, name = word list for r <= 8, else "t%06u"
//   df target : clamp(floor(0.6*N/r), 1, N)  (Zipf alpha = 1)
//   postings  : dense terms (target*8 >= N): one Bernoulli(target/N) sweep over docIds;
//               sparse terms: `target` uniform draws, sorted, de-duplicated  -> docId ascending
//   tf        : 1 + Geometric(p = 0.45), capped at 64 (fixed-point inverse CDF)
//   layout    : 64 barrels, terms_per_barrel = max(1, ceil(V/64)), count == df == realised length
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <exception>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "index_format.hpp"

namespace nsx {

struct Rng {  // xoshiro256** seeded through SplitMix64
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (auto& v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    uint32_t u32() { return (uint32_t)(next() >> 32); }
    uint32_t below(uint32_t n) { return (uint32_t)(((uint64_t)u32() * n) >> 32); }  // multiply-shift
};

inline uint64_t mix_seed(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) {
    uint64_t x = seed ^ (a * 0xD6E8FEB86659FD93ull) ^ (b * 0xA0761D6478BD642Full) ^ (c * 0xE7037ED1A0B428DBull);
    return Rng::splitmix(x);
}

struct GenParams {
    std::string index_dir;
    uint32_t n_segments = 1;
    uint32_t docs_per_segment = 10000;
    uint32_t vocab = 65536;
    uint64_t seed = 1337;       // reference slicer default (scripts/slice_cord19.cpp:25)
    bool legacy_layout = false; // write lexicon.bin + inverted.bin instead of barrels
};

struct GenStats {
    uint64_t total_postings = 0;
    uint64_t total_bytes = 0;
};

inline std::string term_name(uint32_t rank) {
    static const char* words[8] = {"covid", "virus", "vaccine", "infection", "patients", "respiratory", "coronavirus", "pandemic"};
    if (rank >= 1 && rank <= 8) return words[rank - 1];
    char buf[16];
    std::snprintf(buf, sizeof(buf), "t%06u", rank);
    return buf;
}

inline uint32_t df_target(uint32_t N, uint32_t rank) {
    uint64_t t = ((uint64_t)N * 6) / ((uint64_t)10 * rank);
    if (t < 1) t = 1;
    if (t > N) t = N;
    return (uint32_t)t;
}

struct TfSampler {   // P(tf-1 >= k) = 0.55^k in 32.32 fixed point
    uint32_t thr[64];
    TfSampler() {
        uint64_t q = 1ull << 32;          // 0.55^0
        for (int k = 0; k < 64; k++) {
            q = (q * 55) / 100;            // 0.55^(k+1)
            thr[k] = (uint32_t)std::min<uint64_t>(q, 0xFFFFFFFFull);
        }
    }
    uint32_t operator()(uint32_t u) const {   // u uniform u32; tf = 1 + #{k : u < thr[k]} capped at 64
        uint32_t tf = 1;
        for (int k = 0; k < 63 && u < thr[k]; k++) tf++;
        return tf;
    }
};

inline void gen_term_postings(uint32_t N, uint32_t rank, uint64_t seed, uint32_t seg, const TfSampler& tfs,
                              std::vector<uint32_t>& docs, std::vector<uint32_t>& pairs) {
    docs.clear();
    pairs.clear();
    if (N == 0) return;
    Rng rng(mix_seed(seed, seg, rank, 0x7E));
    uint32_t target = df_target(N, rank);
    if ((uint64_t)target * 8 >= N) {
        uint64_t thr = ((uint64_t)target << 32) / N;   // == 2^32 when target == N
        for (uint32_t d = 0; d < N; d++) if ((uint64_t)rng.u32() < thr) docs.push_back(d);
    } else {
        docs.resize(target);
        for (auto& d : docs) d = rng.below(N);
        std::sort(docs.begin(), docs.end());
        docs.erase(std::unique(docs.begin(), docs.end()), docs.end());
    }
    pairs.reserve(docs.size() * 2);
    for (uint32_t d : docs) { pairs.push_back(d); pairs.push_back(tfs(rng.u32())); }
}

inline GenStats generate_index(const GenParams& p) {
    GenStats st;
    fs::path root(p.index_dir);
    fs::create_directories(root / "segments");
    std::vector<std::string> names;
    for (uint32_t seg = 0; seg < p.n_segments; seg++) names.push_back(seg_name(seg + 1));
    const TfSampler tfs;
    std::vector<uint64_t> seg_postings(p.n_segments, 0);
    // segments are independent (every draw is seeded by (seed, segment, rank)): several at a time, same bytes
    auto make_segment = [&](uint32_t seg) {
        const std::string& name = names[seg];
        fs::path d = root / "segments" / name;
        fs::create_directories(d);
        const uint32_t N = p.docs_per_segment;

        // docs.bin + stats.bin
        std::vector<uint32_t> doc_len(N);
        uint64_t total_len = 0;
        {
            Rng rng(mix_seed(p.seed, seg, 0, 0xD0C));
            for (uint32_t i = 0; i < N; i++) {
                uint32_t e = 9 + rng.below(5);
                uint32_t len = (1u << e) + rng.below(1u << e);
                if (rng.below(16) == 0) len = 20 + rng.below(492);
                doc_len[i] = len;
                total_len += len;
            }
            FileOut docs(d / "docs.bin");
            docs.u32(N);
            char uid[24];
            for (uint32_t i = 0; i < N; i++) {
                std::snprintf(uid, sizeof(uid), "u%08u", seg * N + i);
                docs.str(uid);
                docs.str("");
                docs.str("");
                docs.u32(doc_len[i]);
            }
            float avgdl = N ? (float)total_len / (float)N : 0.0f;   // segment_writer.hpp:68 / ForwardIndex.cpp:187
            FileOut stats(d / "stats.bin");
            stats.u32(N);
            stats.f32(avgdl);
        }

        // lexicon + inverted
        const uint32_t V = p.vocab;
        std::vector<uint32_t> docs, pairs;
        if (!p.legacy_layout) {
            uint32_t tpb = (V + kBarrelCount - 1) / kBarrelCount;
            if (tpb == 0) tpb = 1;
            { FileOut bm(d / "barrels.bin"); bm.u32(kBarrelCount); bm.u32(tpb); }
            uint32_t tid = 0;
            for (uint32_t b = 0; b < kBarrelCount; b++) {
                FileOut inv(inv_barrel_path(d, b));
                FileOut lex(lex_barrel_path(d, b));
                lex.u32(0);
                uint32_t nrec = 0;
                uint64_t off = 0;
                // barrel_for_term: min(tid / tpb, 63)  (include/barrels.hpp:42-47)
                while (tid < V && std::min(tid / tpb, kBarrelCount - 1) == b) {
                    gen_term_postings(N, tid + 1, p.seed, seg, tfs, docs, pairs);
                    if (!docs.empty()) {
                        uint32_t df = (uint32_t)docs.size();
                        lex.str(term_name(tid + 1));
                        lex.u32(tid);
                        lex.u32(df);
                        lex.u64(off);
                        lex.u32(df);
                        inv.raw(pairs.data(), pairs.size() * 4);
                        off += (uint64_t)df * 8;
                        nrec++;
                        seg_postings[seg] += df;
                    }
                    tid++;
                }
                lex.patch_u32_at0(nrec);
            }
        } else {
            FileOut inv(d / "inverted.bin");
            FileOut lex(d / "lexicon.bin");
            lex.u32(0);
            uint32_t nrec = 0;
            uint64_t off = 0;
            for (uint32_t tid = 0; tid < V; tid++) {
                gen_term_postings(N, tid + 1, p.seed, seg, tfs, docs, pairs);
                if (docs.empty()) continue;
                uint32_t df = (uint32_t)docs.size();
                lex.str(term_name(tid + 1));
                lex.u32(tid);
                lex.u32(df);
                lex.u64(off);
                lex.u32(df);
                inv.raw(pairs.data(), pairs.size() * 4);
                off += (uint64_t)df * 8;
                nrec++;
                seg_postings[seg] += df;
            }
            lex.patch_u32_at0(nrec);
        }
    };
    {
        const unsigned nt = std::max(1u, std::min<unsigned>({p.n_segments, std::thread::hardware_concurrency(), 16u}));
        if (nt <= 1) {
            for (uint32_t seg = 0; seg < p.n_segments; seg++) make_segment(seg);
        } else {
            std::atomic<uint32_t> next{0};
            std::exception_ptr err;
            std::mutex err_m;
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([&]() {
                    try {
                        for (uint32_t seg = next++; seg < p.n_segments; seg = next++) make_segment(seg);
                    } catch (...) { std::lock_guard<std::mutex> l(err_m); if (!err) err = std::current_exception(); }
                });
            for (auto& t : th) t.join();
            if (err) std::rethrow_exception(err);
        }
    }
    for (uint64_t n : seg_postings) st.total_postings += n;
    save_manifest(root / "manifest.bin", names);
    st.total_bytes = st.total_postings * 8;
    return st;
}

}  // namespace nsx

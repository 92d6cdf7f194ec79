// The reference's `lexicon <SEGMENT_DIR>` tool (src/lexicon.cpp) with the inversion done on the device
// (ns_invert_forward, csrc/ns_invert.hip): reads terms.bin + forward.bin, writes barrels.bin and the 64
// lexicon_bNNN.bin / inverted_bNNN.bin files byte for byte as the reference writes them (:84-146).
#pragma once

#include <chrono>
#include <cstdint>
#include <string>
#include <vector>

#include "index_format.hpp"
#include "nextsearch_hip.h"

namespace nsx {

struct InvertStats {
    uint64_t pairs = 0, kept = 0;
    uint32_t n_docs = 0, n_terms = 0;
    float device_ms = 0.0f;       // HIP events around the device part (expand + radix passes)
    double call_s = 0.0;          // ns_invert_forward, copies included
    double total_s = 0.0;         // files in -> files out
};

inline bool invert_segment(ns_ctx* ctx, const fs::path& seg, InvertStats& st, std::string& err) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    FileBytes tin, fin;
    if (!tin.load(seg / "terms.bin") || !fin.load(seg / "forward.bin")) { err = "Missing forward.bin or terms.bin in: " + seg.string(); return false; }   // :30-33
    const uint32_t tcount = tin.u32();                        // :44
    std::vector<std::string> terms(tcount);
    for (uint32_t i = 0; i < tcount; i++) terms[i] = tin.str();
    // forward.bin: u32 numDocs, then per doc u32 cnt + cnt x {u32 termId, u32 tf} (:60-72)
    const std::vector<uint8_t>& fb = fin.bytes();
    if (fb.size() < 4) { err = "forward.bin is truncated"; return false; }
    const uint32_t n_docs = fin.u32();
    std::vector<uint32_t> counts(n_docs);
    std::vector<uint32_t> pairs;
    pairs.reserve(fb.size() / 4);
    size_t pos = 4;
    for (uint32_t d = 0; d < n_docs; d++) {
        if (pos + 4 > fb.size()) { err = "forward.bin is truncated"; return false; }
        uint32_t cnt;
        std::memcpy(&cnt, fb.data() + pos, 4);
        pos += 4;
        if ((uint64_t)pos + (uint64_t)cnt * 8 > fb.size()) { err = "forward.bin is truncated"; return false; }
        counts[d] = cnt;
        const size_t at = pairs.size();
        pairs.resize(at + (size_t)cnt * 2);
        if (cnt) std::memcpy(pairs.data() + at, fb.data() + pos, (size_t)cnt * 8);
        pos += (size_t)cnt * 8;
    }
    st.n_docs = n_docs; st.n_terms = tcount; st.pairs = pairs.size() / 2;
    std::vector<uint32_t> df(tcount);
    std::vector<uint8_t> postings(pairs.size() * 4);
    const auto t1 = clk::now();
    int rc = ns_invert_forward(ctx, counts.data(), n_docs, pairs.data(), st.pairs, tcount, df.data(), postings.data(), &st.kept, &st.device_ms);
    st.call_s = std::chrono::duration<double>(clk::now() - t1).count();
    if (rc != NS_OK) { err = std::string("ns_invert_forward: ") + ns_last_error(ctx); return false; }

    // barrels (:84-146): 64 of them, ceil(tcount / 64) consecutive termIds each
    const uint32_t barrel_count = 64;
    uint32_t tpb = (tcount + barrel_count - 1) / barrel_count;
    if (tpb == 0) tpb = 1;
    try {
        { FileOut m(seg / "barrels.bin"); m.u32(barrel_count); m.u32(tpb); }
        uint64_t src = 0;   // byte position in `postings` (lists are in termId order == barrel order)
        uint32_t tid = 0;
        for (uint32_t b = 0; b < barrel_count; b++) {
            FileOut inv(inv_barrel_path(seg, b)), lex(lex_barrel_path(seg, b));
            lex.u32(0);
            uint32_t in_barrel = 0;
            uint64_t off = 0;
            const uint64_t first = src;
            // barrel_for_term: tid / tpb, the last barrel takes the rest (include/barrels.hpp:43-48)
            for (; tid < tcount && (b + 1 == barrel_count || tid / tpb == b); tid++) {
                const uint32_t n = df[tid];
                if (!n) continue;                                    // :108
                in_barrel++;
                lex.str(terms[tid]); lex.u32(tid); lex.u32(n); lex.u64(off); lex.u32(n);   // :117-121
                off += (uint64_t)n * 8;
                src += (uint64_t)n * 8;
            }
            inv.raw(postings.data() + first, (size_t)(src - first));
            lex.patch_u32_at0(in_barrel);                              // :133-146
        }
    } catch (const std::exception& ex) { err = ex.what(); return false; }
    st.total_s = std::chrono::duration<double>(clk::now() - t0).count();
    return true;
}

}  // namespace nsx

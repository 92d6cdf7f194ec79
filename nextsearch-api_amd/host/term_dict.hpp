// Query preparation at batch rate: term -> per-segment {byte_off, count, idf} in ONE probe.
//
// The reference probes `seg.lex.find(term)` — a std::unordered_map<std::string, LexEntry> — once per (term, segment)
// and evaluates bm25_idf behind it (src/api_engine.cpp:454-461).  The loaded lexicons stay what they are
// (index_format.hpp: the load path is unchanged); at reload() this table is built NEXT TO them:
//   * one dictionary over the terms of all segments: open addressing, 64-bit FNV-1a over the term's bytes, the
//     bytes themselves in one pool (a probe touches one table line and one pool line, no std::string);
//   * per (term, segment) the three numbers a term ref needs, precomputed: the list's byte offset inside the
//     segment's payload, LexEntry.count, and bm25_idf(N, df) — the host's glibc logf value (api_engine.cpp:45-47),
//     bit for bit what Engine::build_refs_range computed per probe before.
// A (term, segment) pair the reference would skip (:455 absent, :458 df == 0) carries byte_off == kAbsent.
// The tokenizer below restates include/textutil.hpp:13-37 + src/api_engine.cpp:391-397 over raw bytes without
// allocating (same tokens, same filter, same order and duplicates as textutil.hpp's std::string form, which stays
// the authority the CPU tests compare against).
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

#include "index_format.hpp"

namespace nsx {

struct TermSeg {
    uint64_t byte_off;
    uint32_t count;
    float idf;
};
static constexpr uint64_t kAbsent = ~0ull;

inline uint64_t term_hash(const char* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
    return h ^ (h >> 29);
}

class TermDict {
public:
    void clear() { slots_.clear(); ents_.clear(); pool_.clear(); per_seg_.clear(); n_segs_ = 0; mask_ = 0; }
    bool empty() const { return ents_.empty(); }
    uint32_t n_segments() const { return n_segs_; }
    size_t n_terms() const { return ents_.size(); }

    // idf_of(N, df): the engine's bm25_idf (passed in so that this header does not depend on engine.hpp)
    template <class IdfFn>
    void build(const std::vector<SegmentData>& segs, IdfFn idf_of) {
        clear();
        n_segs_ = (uint32_t)segs.size();
        size_t total = 0;
        for (const auto& s : segs) total += s.lex.size();
        size_t cap = 16;
        while (cap < total * 2 + 2) cap <<= 1;   // load factor <= 0.5 even when no term is shared between segments
        slots_.assign(cap, 0u);
        mask_ = cap - 1;
        ents_.reserve(total);
        for (uint32_t sid = 0; sid < n_segs_; sid++) {
            const SegmentData& s = segs[sid];
            for (const auto& kv : s.lex) {
                const uint32_t gid = intern(kv.first);
                if (per_seg_.size() < (size_t)(gid + 1) * n_segs_) per_seg_.resize((size_t)(gid + 1) * n_segs_, TermSeg{kAbsent, 0u, 0.0f});
                const LexEntry& e = kv.second;
                if (e.df == 0) continue;                                        // src/api_engine.cpp:458
                if (s.use_barrels && e.barrelId >= s.barrel_base.size()) continue;   // cannot be located: treated as absent
                per_seg_[(size_t)gid * n_segs_ + sid] = TermSeg{s.list_byte_offset(e), e.count, idf_of(s.N, e.df)};
            }
        }
        per_seg_.resize(ents_.size() * (size_t)n_segs_, TermSeg{kAbsent, 0u, 0.0f});
    }

    // index of the term, or -1
    int64_t find(const char* p, size_t n) const {
        if (ents_.empty()) return -1;
        const uint64_t h = term_hash(p, n);
        const uint32_t tag = (uint32_t)(h >> 32) | 1u;
        for (size_t i = (size_t)h & mask_;; i = (i + 1) & mask_) {
            const uint64_t sl = slots_[i];
            if (sl == 0) return -1;
            if ((uint32_t)(sl >> 32) != tag) continue;
            const Ent& e = ents_[(uint32_t)sl - 1u];
            if (e.len == n && std::memcmp(pool_.data() + e.off, p, n) == 0) return (int64_t)((uint32_t)sl - 1u);
        }
    }
    const TermSeg* row(uint32_t gid) const { return per_seg_.data() + (size_t)gid * n_segs_; }

private:
    struct Ent { uint32_t off, len; };
    uint32_t intern(const std::string& t) {
        const uint64_t h = term_hash(t.data(), t.size());
        const uint32_t tag = (uint32_t)(h >> 32) | 1u;
        for (size_t i = (size_t)h & mask_;; i = (i + 1) & mask_) {
            const uint64_t sl = slots_[i];
            if (sl == 0) {
                ents_.push_back(Ent{(uint32_t)pool_.size(), (uint32_t)t.size()});
                pool_.insert(pool_.end(), t.begin(), t.end());
                slots_[i] = ((uint64_t)tag << 32) | (uint64_t)ents_.size();   // index + 1: 0 stays "empty"
                return (uint32_t)ents_.size() - 1u;
            }
            if ((uint32_t)(sl >> 32) != tag) continue;
            const Ent& e = ents_[(uint32_t)sl - 1u];
            if (e.len == t.size() && std::memcmp(pool_.data() + e.off, t.data(), t.size()) == 0) return (uint32_t)sl - 1u;
        }
    }
    std::vector<uint64_t> slots_;    // tag << 32 | (entry index + 1); 0 = empty
    std::vector<Ent> ents_;
    std::vector<char> pool_;
    std::vector<TermSeg> per_seg_;   // [term][segment]
    uint32_t n_segs_ = 0;
    size_t mask_ = 0;
};

// include/textutil.hpp:31-37 over a lowercase byte span
inline bool is_stopword_span(const char* p, size_t n) {
    switch (n) {
        case 1: return p[0] == 'a';
        case 2: {
            const uint16_t w = (uint16_t)((unsigned char)p[0] | ((unsigned char)p[1] << 8));
            auto k = [](char a, char b) { return (uint16_t)((unsigned char)a | ((unsigned char)b << 8)); };
            return w == k('a', 'n') || w == k('o', 'r') || w == k('o', 'f') || w == k('t', 'o') || w == k('i', 'n') || w == k('o', 'n') ||
                   w == k('b', 'y') || w == k('a', 's') || w == k('i', 's') || w == k('b', 'e') || w == k('i', 't') || w == k('a', 't');
        }
        case 3: return !std::memcmp(p, "the", 3) || !std::memcmp(p, "and", 3) || !std::memcmp(p, "for", 3) || !std::memcmp(p, "are", 3) || !std::memcmp(p, "was", 3);
        case 4: return !std::memcmp(p, "with", 4) || !std::memcmp(p, "were", 4) || !std::memcmp(p, "been", 4) || !std::memcmp(p, "this", 4) ||
                       !std::memcmp(p, "that", 4) || !std::memcmp(p, "from", 4);
        default: return false;
    }
}

// Calls fn(ptr, len) for every base term of `text` (src/api_engine.cpp:391-397: tokens of include/textutil.hpp:13-28
// with size() >= 2 that are no stop-words), in order, duplicates included.  `scratch` holds the lowercased bytes.
template <class Fn>
inline void for_each_base_term(const char* text, size_t n, std::vector<char>& scratch, Fn fn) {
    if (scratch.size() < n + 1) scratch.resize(n + 1);
    char* buf = scratch.data();
    size_t len = 0;
    for (size_t i = 0; i <= n; i++) {
        const unsigned char c = i < n ? (unsigned char)text[i] : 0;
        const bool alnum = (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
        if (alnum) { buf[len++] = (c >= 'A' && c <= 'Z') ? (char)(c - 'A' + 'a') : (char)c; continue; }
        if (len >= 2 && !is_stopword_span(buf, len)) fn((const char*)buf, len);
        len = 0;
    }
}

}  // namespace nsx

// Semantic query expansion around the device similarity search.
// Host mirror of cord19::SemanticIndex (include/semantic_embedding.hpp, src/semantic_embedding.cpp): the text
// loader and the weighting / merging rules of expand() run here, most_similar_to_vec runs on the device
// (ns_sem_topk) for ALL query vectors of a batch in one call per top-k size.
//
// What has to match the reference for the hot path's results to stay bit-identical:
//   * the table: rows in file order, only needed terms, first dim wins, < 10 values skipped, L2 norm with a
//     double sum and (float)(x / n) (:18-24, :61-95) — parsed with the same stream extraction;
//   * sims: sequential fp32 dot (device), neighbours filtered by min_sim and the banned (query) rows;
//   * weights: max(0, min(alpha, alpha*sim)) per term neighbour (:186), alpha*0.8f for the centroid's (:213);
//     a candidate keeps its LARGEST weight (:188, :215); original terms weigh 1.0 (:160);
//   * ORDER of the weighted terms — it is the fp32 accumulation order of the scores: the reference collects
//     them in a std::unordered_map (reserve(2 * max_total_terms), :156-157), copies the map's iteration
//     order into a vector and std::sorts it by weight (:222-228).  The same container, the same sequence
//     of insertions and the same sort give the same order with the same libstdc++; nothing else would.
#pragma once

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <fstream>
#include <sstream>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "index_format.hpp"
#include "nextsearch_hip.h"

namespace nsx {

using WeightedTerms = std::vector<std::pair<std::string, float>>;

// Reads numbers and words off a span of text the way the stream extractors the reference uses do in the "C" locale:
// leading white space is skipped, then the longest prefix the type's grammar admits is taken and converted; a failed
// extraction ends the sequence (the reference's `while (iss >> x)`).
struct TextCursor {
    const char* p;
    const char* end;
    static bool blank(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
    static bool digit(char c) { return c >= '0' && c <= '9'; }
    void skip_blanks() { while (p < end && blank(*p)) p++; }
    bool only_blanks_left() { skip_blanks(); return p == end; }
    bool token(std::string_view& out) {
        skip_blanks();
        const char* b = p;
        while (p < end && !blank(*p)) p++;
        out = std::string_view(b, (size_t)(p - b));
        return p != b;
    }
    // [+-]? digits; out-of-range fails (num_get sets failbit)
    bool integer(long long& out) {
        skip_blanks();
        const char* b = p;
        if (p < end && (*p == '+' || *p == '-')) p++;
        const char* d = p;
        while (p < end && digit(*p)) p++;
        if (p == d) return false;
        const std::string tmp(b, p);
        errno = 0;
        char* stop = nullptr;
        out = std::strtoll(tmp.c_str(), &stop, 10);
        return errno != ERANGE && stop && *stop == 0;
    }
    // libstdc++'s float grammar: [+-]? then digits with at most one '.', then — only after a digit — one e/E with an
    // optional sign, then digits.  The prefix goes through strtof; unconverted characters or +-HUGE_VALF fail.
    bool real(float& out) {
        skip_blanks();
        const char* b = p;
        if (p < end && (*p == '+' || *p == '-')) p++;
        bool mantissa = false, point = false, expo = false;
        for (; p < end; p++) {
            const char c = *p;
            if (digit(c)) { mantissa = true; continue; }
            if (c == '.' && !point && !expo) { point = true; continue; }
            if ((c == 'e' || c == 'E') && !expo && mantissa) {
                expo = true;
                if (p + 1 < end && (p[1] == '+' || p[1] == '-')) p++;
                continue;
            }
            break;
        }
        if (p == b) return false;
        const std::string tmp(b, p);
        char* stop = nullptr;
        const float v = std::strtof(tmp.c_str(), &stop);
        if (!stop || stop == tmp.c_str() || *stop != 0 || std::isinf(v)) return false;
        out = v;
        return true;
    }
};

class SemanticTable {
public:
    bool enabled = false;
    int dim = 0;
    std::vector<std::string> terms;          // row -> term
    std::vector<float> vecs;                 // row-major, L2-normalised
    std::unordered_map<std::string, uint32_t> term_to_row;
    ns_sem* dev = nullptr;                   // the table on the device (owned by the Engine's ctx)

    // expand() parameters as the engine passes them (src/api_engine.cpp:412-417)
    int per_term = 3, global_topk = 5, max_total_terms = 40;
    float min_sim = 0.55f, alpha = 0.6f;

    void clear() { enabled = false; dim = 0; terms.clear(); vecs.clear(); term_to_row.clear(); }

    // v / |v| with the norm summed in double and each component rounded once from the double quotient
    // (src/semantic_embedding.cpp:18-24); a zero vector stays as it is.
    static void to_unit_length(float* v, size_t n) {
        double sq = 0.0;
        for (size_t i = 0; i < n; i++) sq += (double)v[i] * (double)v[i];
        const double len = std::sqrt(sq);
        if (!(len > 0.0)) return;
        for (size_t i = 0; i < n; i++) v[i] = (float)((double)v[i] / len);
    }

    // The embeddings text file (src/semantic_embedding.cpp:35-101): per non-empty line a word and its values; an
    // optional first line "<count> <dim>" (two positive integers, dim < 5000, nothing else) is a header.  Kept: words in
    // `needed` (all words when `needed` is empty) with >= 10 values and the dimension of the first kept line; a
    // repeated word gets a row of its own but resolves by name to its first row.  The reference reads the lines with
    // stream extraction; TextCursor below applies the same extraction rules to the file's bytes in place.
    bool load_from_text(const fs::path& path, const std::unordered_set<std::string>& needed) {
        clear();
        FileBytes file;
        if (!fs::exists(path) || !file.load(path)) return false;
        const std::string_view text((const char*)file.bytes().data(), file.size());
        std::vector<float> row;
        std::string word;
        bool header_possible = true;
        for (size_t pos = 0; pos < text.size();) {
            size_t eol = text.find('\n', pos);
            if (eol == std::string_view::npos) eol = text.size();
            TextCursor line{text.data() + pos, text.data() + eol};
            pos = eol + 1;
            if (line.p == line.end) continue;                       // empty line: not even a header candidate
            if (header_possible) {
                header_possible = false;
                TextCursor h = line;
                long long count = 0, width = 0;
                if (h.integer(count) && h.integer(width) && h.only_blanks_left() && count > 0 && width > 0 && width < 5000) continue;
            }
            std::string_view w;
            if (!line.token(w)) continue;
            word.assign(w);
            if (!needed.empty() && !needed.count(word)) continue;
            row.clear();
            for (float x; line.real(x);) row.push_back(x);          // stops at the first thing that is not a number
            if (row.size() < 10) continue;
            if (dim == 0) dim = (int)row.size();
            if ((int)row.size() != dim) continue;
            to_unit_length(row.data(), row.size());
            term_to_row.emplace(word, (uint32_t)terms.size());       // emplace: the first row of a word keeps the name
            terms.push_back(word);
            vecs.insert(vecs.end(), row.begin(), row.end());
        }
        enabled = !terms.empty() && dim > 0;
        return enabled;
    }

    const float* vec_of(const std::string& t) const {
        auto it = term_to_row.find(t);
        return it == term_to_row.end() ? nullptr : &vecs[(size_t)it->second * (size_t)dim];
    }

    // expand() for a batch of queries: out[q] = the weighted terms of query q in the reference's order.
    // Queries with no terms get an empty list.  false + err on a device error.
    bool expand_batch(ns_ctx* ctx, const std::vector<std::vector<std::string>>& qterms, std::vector<WeightedTerms>& out, std::string& err) const {
        const size_t Q = qterms.size();
        out.assign(Q, {});
        if (!enabled || dim <= 0 || !dev) { err = "semantic table not loaded"; return false; }
        // one query vector per (query, term occurrence with a vector), one centroid per query that has any
        std::vector<float> tv, cv;
        std::vector<uint32_t> t_ban_off{0}, c_ban_off{0}, ban_rows_t, ban_rows_c;
        std::vector<uint32_t> t_first(Q + 1, 0), c_index(Q, 0xFFFFFFFFu);
        for (size_t q = 0; q < Q; q++) {
            const auto& qt = qterms[q];
            t_first[q] = (uint32_t)(t_ban_off.size() - 1);
            std::unordered_set<uint32_t> banned;
            for (const auto& t : qt) { auto it = term_to_row.find(t); if (it != term_to_row.end()) banned.insert(it->second); }
            std::vector<uint32_t> bl(banned.begin(), banned.end());
            std::vector<float> cen((size_t)dim, 0.0f);
            int cnt = 0;
            for (const auto& t : qt) {
                const float* v = vec_of(t);
                if (!v) continue;
                tv.insert(tv.end(), v, v + dim);
                ban_rows_t.insert(ban_rows_t.end(), bl.begin(), bl.end());
                t_ban_off.push_back((uint32_t)ban_rows_t.size());
                for (int j = 0; j < dim; j++) cen[(size_t)j] += v[j];          // :197
                cnt++;
            }
            if (global_topk > 0 && cnt > 0) {
                for (int j = 0; j < dim; j++) cen[(size_t)j] /= (float)cnt;     // :202
                to_unit_length(cen.data(), cen.size());
                c_index[q] = (uint32_t)(c_ban_off.size() - 1);
                cv.insert(cv.end(), cen.begin(), cen.end());
                ban_rows_c.insert(ban_rows_c.end(), bl.begin(), bl.end());
                c_ban_off.push_back((uint32_t)ban_rows_c.size());
            }
        }
        t_first[Q] = (uint32_t)(t_ban_off.size() - 1);
        const uint32_t nt = t_first[Q], nc = (uint32_t)(c_ban_off.size() - 1);
        std::vector<uint32_t> t_rows((size_t)nt * std::max(per_term, 1)), t_cnt(nt), c_rows((size_t)nc * std::max(global_topk, 1)), c_cnt(nc);
        std::vector<float> t_sims(t_rows.size()), c_sims(c_rows.size());
        if (nt && per_term > 0 &&
            ns_sem_topk(ctx, dev, tv.data(), nt, (uint32_t)per_term, min_sim, t_ban_off.data(), ban_rows_t.data(), t_rows.data(), t_sims.data(), t_cnt.data(), nullptr) != NS_OK) {
            err = std::string("ns_sem_topk: ") + ns_last_error(ctx);
            return false;
        }
        if (nc && ns_sem_topk(ctx, dev, cv.data(), nc, (uint32_t)global_topk, min_sim, c_ban_off.data(), ban_rows_c.data(), c_rows.data(), c_sims.data(), c_cnt.data(), nullptr) != NS_OK) {
            err = std::string("ns_sem_topk: ") + ns_last_error(ctx);
            return false;
        }
        for (size_t q = 0; q < Q; q++) {
            const auto& qt = qterms[q];
            if (qt.empty()) continue;
            // weights by term; the ORDER of first insertion decides the map's iteration order, hence the scoring order
            std::unordered_map<std::string, float> w;
            w.reserve((size_t)max_total_terms * 2);
            for (const auto& t : qt) if (!t.empty()) w[t] = 1.0f;                    // the query's own terms (:160)
            // a neighbour weighs cap * sim, at most cap, never negative, and keeps the largest weight it was offered (:186-188, :213-215)
            auto offer = [&](const uint32_t* rows, const float* sims, uint32_t n, float cap) {
                for (uint32_t j = 0; j < n; j++) {
                    const float weight = std::max(0.0f, std::min(cap, cap * sims[j]));
                    const auto [slot, fresh] = w.try_emplace(terms[rows[j]], weight);
                    if (!fresh && weight > slot->second) slot->second = weight;
                }
            };
            for (uint32_t i = t_first[q]; per_term > 0 && i < t_first[q + 1]; i++)
                offer(&t_rows[(size_t)i * per_term], &t_sims[(size_t)i * per_term], t_cnt[i], alpha);
            if (const uint32_t i = c_index[q]; i != 0xFFFFFFFFu)
                offer(&c_rows[(size_t)i * global_topk], &c_sims[(size_t)i * global_topk], c_cnt[i], alpha * 0.8f);
            WeightedTerms& o = out[q];
            o.reserve(w.size());
            for (auto& kv : w) o.push_back(kv);
            std::sort(o.begin(), o.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
            if ((int)o.size() > max_total_terms) o.resize((size_t)max_total_terms);
        }
        return true;
    }
};

}  // namespace nsx

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
#include <cmath>
#include <cstdint>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "index_format.hpp"
#include "nextsearch_hip.h"

namespace nsx {

using WeightedTerms = std::vector<std::pair<std::string, float>>;

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

    static void l2_normalize(std::vector<float>& v) {
        double ss = 0.0;
        for (float x : v) ss += (double)x * (double)x;
        const double n = std::sqrt(ss);
        if (n <= 0.0) return;
        for (float& x : v) x = (float)(x / n);
    }

    bool load_from_text(const fs::path& path, const std::unordered_set<std::string>& needed) {
        clear();
        std::ifstream in(path);
        if (!in.is_open()) return false;
        std::string line;
        bool first = true;
        size_t loaded = 0;
        while (std::getline(in, line)) {
            if (line.empty()) continue;
            if (first) {                                  // optional "<vocab> <dim>" header (:52-59, :66-69)
                first = false;
                std::istringstream h(line);
                long long a, b;
                std::string extra;
                if ((h >> a >> b) && !(h >> extra) && a > 0 && b > 0 && b < 5000) continue;
            }
            std::istringstream iss(line);
            std::string word;
            if (!(iss >> word)) continue;
            if (!needed.empty() && needed.find(word) == needed.end()) continue;
            std::vector<float> v;
            float x;
            while (iss >> x) v.push_back(x);
            if (v.size() < 10) continue;
            if (dim == 0) dim = (int)v.size();
            if ((int)v.size() != dim) continue;
            l2_normalize(v);
            const uint32_t row = (uint32_t)terms.size();
            terms.push_back(word);
            term_to_row.emplace(word, row);               // a repeated word keeps its FIRST row here, yet gets a new row
            vecs.insert(vecs.end(), v.begin(), v.end());
            loaded++;
        }
        enabled = loaded > 0 && dim > 0;
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
                l2_normalize(cen);
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
            std::unordered_map<std::string, float> w;
            w.reserve((size_t)max_total_terms * 2);
            for (const auto& t : qt) if (!t.empty()) w[t] = 1.0f;
            if (per_term > 0)
                for (uint32_t i = t_first[q]; i < t_first[q + 1]; i++)
                    for (uint32_t j = 0; j < t_cnt[i]; j++) {
                        const std::string& cand = terms[t_rows[(size_t)i * per_term + j]];
                        const float weight = std::max(0.0f, std::min(alpha, alpha * t_sims[(size_t)i * per_term + j]));
                        auto it = w.find(cand);
                        if (it == w.end() || weight > it->second) w[cand] = weight;
                    }
            if (c_index[q] != 0xFFFFFFFFu) {
                const uint32_t i = c_index[q];
                for (uint32_t j = 0; j < c_cnt[i]; j++) {
                    const std::string& cand = terms[c_rows[(size_t)i * global_topk + j]];
                    const float weight = std::max(0.0f, std::min(alpha * 0.8f, alpha * 0.8f * c_sims[(size_t)i * global_topk + j]));
                    auto it = w.find(cand);
                    if (it == w.end() || weight > it->second) w[cand] = weight;
                }
            }
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

// ns_tool — small CLI over the host facade.
//   ns_tool gen-index <index_dir> <n_segments> <docs_per_segment> [vocab=65536] [seed=1337] [--legacy]
//   ns_tool search <index_dir> <k> <query text ...>        (needs an MI355X; prints the /api/search JSON body)
//   ns_tool facade-bench <index_dir> <queries.txt> <k> [reps=5] [device=0]
//        times the C++ facade from INSIDE the process (no ctypes, no Python): query preparation alone (tokenise,
//        dictionary probes, idf: src/api_engine.cpp:388-397,:454-461) and Engine::search_batch_flat, query TEXT in ->
//        hits in host memory out.  device < 0: query preparation only (runs without a GPU).  One JSON line.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <fstream>
#include <vector>
#include <algorithm>

#include "engine.hpp"
#include "gen_index.hpp"

int main(int argc, char** argv) {
    if (argc >= 5 && std::strcmp(argv[1], "gen-index") == 0) {
        nsx::GenParams p;
        p.index_dir = argv[2];
        p.n_segments = (uint32_t)std::strtoul(argv[3], nullptr, 10);
        p.docs_per_segment = (uint32_t)std::strtoul(argv[4], nullptr, 10);
        int pos = 0;
        for (int i = 5; i < argc; i++) {
            if (std::strcmp(argv[i], "--legacy") == 0) { p.legacy_layout = true; continue; }
            if (pos == 0) p.vocab = (uint32_t)std::strtoul(argv[i], nullptr, 10);
            if (pos == 1) p.seed = std::strtoull(argv[i], nullptr, 10);
            pos++;
        }
        nsx::GenStats st = nsx::generate_index(p);
        std::printf("{\"postings\": %llu, \"bytes\": %llu}\n", (unsigned long long)st.total_postings, (unsigned long long)st.total_bytes);
        return 0;
    }
    if (argc >= 5 && std::strcmp(argv[1], "search") == 0) {
        nextsearch::Engine eng(0);
        eng.index_dir = argv[2];
        if (!eng.reload()) { std::fprintf(stderr, "reload failed: %s\n", eng.last_error().c_str()); return 1; }
        int k = std::atoi(argv[3]);
        std::string q;
        for (int i = 4; i < argc; i++) { if (i > 4) q.push_back(' '); q += argv[i]; }
        std::printf("%s\n", eng.search(q, k).c_str());
        return 0;
    }
    if (argc >= 5 && std::strcmp(argv[1], "facade-bench") == 0) {
        const int k = std::atoi(argv[4]);
        const int reps = argc > 5 ? std::max(1, std::atoi(argv[5])) : 5;
        const int device = argc > 6 ? std::atoi(argv[6]) : 0;
        std::vector<std::string> qs;
        {
            std::ifstream in(argv[3]);
            std::string ln;
            while (std::getline(in, ln)) qs.push_back(ln);
        }
        if (qs.empty()) { std::fprintf(stderr, "no queries in %s\n", argv[3]); return 1; }
        nextsearch::Engine eng(device);
        eng.index_dir = argv[2];
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto r0 = now();
        if (!eng.reload()) { std::fprintf(stderr, "reload failed: %s\n", eng.last_error().c_str()); return 1; }
        const double reload_ms = ms(r0, now());
        const size_t Q = qs.size();
        const int K = std::max(1, std::min(k, 100));
        std::vector<nextsearch::Engine::QueryView> views(Q);
        for (size_t q = 0; q < Q; q++) views[q] = {qs[q].data(), qs[q].size()};
        std::vector<ns_query_desc> qd;
        std::vector<ns_term_ref> refs;
        std::vector<uint8_t> usable;
        std::vector<double> t_prep, t_flat;
        for (int r = 0; r < reps + 1; r++) {
            const auto a = now();
            eng.build_refs(qs, qd, refs, usable);
            if (r) t_prep.push_back(ms(a, now()));
        }
        std::vector<ns_hit> hits(Q * (size_t)K);
        std::vector<uint32_t> nhits(Q);
        std::vector<uint64_t> found(Q);
        uint64_t check = 0;
        if (device >= 0) {
            for (int r = 0; r < reps + 2; r++) {
                const auto a = now();
                if (!eng.search_batch_flat(views.data(), Q, k, NS_FLAG_OR, hits.data(), nhits.data(), found.data(), usable.data())) {
                    std::fprintf(stderr, "search_batch_flat failed: %s\n", eng.last_error().c_str());
                    return 1;
                }
                if (r >= 2) t_flat.push_back(ms(a, now()));
            }
            for (size_t q = 0; q < Q; q++) check += found[q] + nhits[q];
        }
        auto med = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        const double p = med(t_prep), f = med(t_flat);
        std::printf("{\"queries\": %zu, \"k\": %d, \"term_refs\": %zu, \"reload_ms\": %.1f, \"query_prep_ms\": %.3f, \"query_prep_qps\": %.0f, "
                    "\"search_batch_flat_ms\": %.3f, \"search_batch_flat_qps\": %.0f, \"reps\": %d, \"checksum\": %llu}\n",
                    Q, K, refs.size(), reload_ms, p, p > 0 ? Q / (p * 1e-3) : 0.0, f, f > 0 ? Q / (f * 1e-3) : 0.0, reps, (unsigned long long)check);
        return 0;
    }
    std::fprintf(stderr, "usage: %s gen-index <dir> <n_segments> <docs_per_segment> [vocab] [seed] [--legacy]\n       %s search <dir> <k> <query...>\n", argv[0], argv[0]);
    return 2;
}

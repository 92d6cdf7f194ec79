// TEST INFRASTRUCTURE ONLY — driver around the *real* reference engine.
//
// This file is ours; it #includes the reference's own headers and is linked against the
// reference's own translation units compiled where they lie under /root/reference (see
// oracle/Makefile, target `ref`).  No reference source is copied into this repository and the
// resulting binary (oracle/_ref/ref_driver) is git-ignored.  It exists to
//   (1) capture golden vectors from cord19::Engine::search (src/api_engine.cpp:369-542) that pin
//       the C restatement in oracle/bm25_oracle.c and, through it, the HIP path;
//   (2) serve as the `cpu_baseline.kind == "reference"` leg of bench.py (it travels to the GPU box
//       as a prebuilt binary; /root/reference itself never does).
//
// Usage:
//   ref_driver search <index_dir> <queries.txt> <K> <out.txt>
//        one query per line; writes per query:  "Q <found|-1> <nhits>" then nhits lines
//        "<segIdx> <docId> <score-bits-hex>"
//   ref_driver json <index_dir> <queries.txt> <K> <out.txt>
//        writes per query "J <number of bytes>\n" followed by exactly that many bytes of
//        Engine::search(...).dump(2) and a newline: the reference's own result assembly
//        (src/api_engine.cpp:400-404,:505-536) incl. the metadata.csv decoration
//   ref_driver time <index_dir> <queries.txt> <K> <max_seconds>
//        prints one JSON line {"queries":n,"seconds":s,"qps":...}; the search-result cache is
//        emptied after every call so search_cache.json rewrites stay O(1) (api_engine.cpp:245-249).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <unistd.h>
#include <vector>

#include "api_engine.hpp"
#include "api_segment.hpp"
#include "textutil.hpp"

using cord19::json;

static std::vector<std::string> read_lines(const std::string& path) {
    std::vector<std::string> out;
    std::ifstream in(path);
    std::string line;
    while (std::getline(in, line)) out.push_back(line);
    return out;
}

static void drop_cache(cord19::Engine& e) {
    e.cache.clear();
    e.lru_list.clear();
    e.cache_updates_since_save = 0;
}

int main(int argc, char** argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s search|json|expand|semtable|time <index_dir> <queries.txt> <K> <out|max_seconds>\n", argv[0]);
        return 2;
    }
    std::string mode = argv[1];
    std::string index_dir = fs::absolute(argv[2]).string();
    std::string qpath = fs::absolute(argv[3]).string();
    int K = std::atoi(argv[4]);
    std::string last = argv[5];
    std::string outpath = (mode == "search" || mode == "json" || mode == "json2" || mode == "expand" || mode == "semtable") ? fs::absolute(last).string() : std::string();

    auto queries = read_lines(qpath);

    // The engine persists JSON caches into the CWD: run from a private scratch directory.
    char tmpl[] = "/tmp/ns_ref_scratch_XXXXXX";
    char* scratch = mkdtemp(tmpl);
    if (!scratch || chdir(scratch) != 0) { std::perror("scratch"); return 1; }

    int rc = 0;
    {
        cord19::Engine engine;
        engine.index_dir = index_dir;
        if (!engine.reload()) { std::fprintf(stderr, "reload failed for %s\n", index_dir.c_str()); return 1; }
        drop_cache(engine);

        std::unordered_map<std::string, uint32_t> seg_index;
        for (uint32_t i = 0; i < engine.seg_names.size(); i++) seg_index[engine.seg_names[i]] = i;

        if (mode == "search") {
            std::FILE* out = std::fopen(outpath.c_str(), "w");
            if (!out) { std::perror("out"); return 1; }
            for (auto& q : queries) {
                json j = engine.search(q, K);
                drop_cache(engine);
                long long found = -1;
                if (j.count("found")) found = (long long)j["found"].get<uint64_t>();
                auto& res = j["results"];
                std::fprintf(out, "Q %lld %zu\n", found, res.size());
                for (auto& r : res) {
                    float s = (float)r["score"].get<double>();
                    uint32_t bits;
                    std::memcpy(&bits, &s, 4);
                    uint32_t seg = seg_index.at(r["segment"].get<std::string>());
                    uint32_t doc = r["docId"].get<uint32_t>();
                    std::fprintf(out, "%u %u %08x\n", seg, doc, bits);
                }
            }
            std::fclose(out);
        } else if (mode == "json") {
            std::FILE* out = std::fopen(outpath.c_str(), "w");
            if (!out) { std::perror("out"); return 1; }
            for (auto& q : queries) {
                json j = engine.search(q, K);
                drop_cache(engine);
                const std::string text = j.dump(2);
                std::fprintf(out, "J %zu\n", text.size());
                std::fwrite(text.data(), 1, text.size(), out);
                std::fputc('\n', out);
            }
            std::fclose(out);
        } else if (mode == "json2") {
            // the search-result cache (src/api_engine.cpp:190-250): the SECOND answer to each query, cache kept
            std::FILE* out = std::fopen(outpath.c_str(), "w");
            if (!out) { std::perror("out"); return 1; }
            for (auto& q : queries) {
                (void)engine.search(q, K);
                json j = engine.search(q, K);
                const std::string text = j.dump(2);
                std::fprintf(out, "J %zu\n", text.size());
                std::fwrite(text.data(), 1, text.size(), out);
                std::fputc('\n', out);
            }
            std::fprintf(out, "C %zu\n", engine.cache.size());
            std::fclose(out);
            drop_cache(engine);
        } else if (mode == "expand") {
            // the weighted terms Engine::search scores (src/api_engine.cpp:386-417), in scoring order
            std::FILE* out = std::fopen(outpath.c_str(), "w");
            if (!out) { std::perror("out"); return 1; }
            std::fprintf(out, "S %d %zu %d\n", engine.sem.enabled ? 1 : 0, engine.sem.terms.size(), engine.sem.dim);
            for (auto& q : queries) {
                auto qtoks = tokenize(q);
                std::vector<std::string> base_terms;
                for (auto& t : qtoks) {
                    if (t.size() < 2) continue;
                    if (is_stopword(t)) continue;
                    base_terms.push_back(t);
                }
                std::vector<std::pair<std::string, float>> w;
                if (!base_terms.empty() && engine.sem.enabled) w = engine.sem.expand(base_terms, 3, 5, 0.55f, 0.6f, 40);
                std::fprintf(out, "E %zu\n", w.size());
                for (auto& tw : w) {
                    uint32_t bits;
                    std::memcpy(&bits, &tw.second, 4);
                    std::fprintf(out, "%s\t%08x\n", tw.first.c_str(), bits);
                }
            }
            std::fclose(out);
        } else if (mode == "semtable") {
            // the embedding table as SemanticIndex::load_from_text left it (src/semantic_embedding.cpp:35-101): per row the
            // term and the fp32 bits of its normalised vector
            std::FILE* out = std::fopen(outpath.c_str(), "w");
            if (!out) { std::perror("out"); return 1; }
            std::fprintf(out, "S %d %zu %d\n", engine.sem.enabled ? 1 : 0, engine.sem.terms.size(), engine.sem.dim);
            for (size_t r = 0; r < engine.sem.terms.size(); r++) {
                std::fprintf(out, "%s", engine.sem.terms[r].c_str());
                for (int j = 0; j < engine.sem.dim; j++) {
                    uint32_t bits;
                    std::memcpy(&bits, &engine.sem.vecs[r * (size_t)engine.sem.dim + j], 4);
                    std::fprintf(out, " %08x", bits);
                }
                std::fprintf(out, "\n");
            }
            std::fclose(out);
        } else if (mode == "time") {
            double max_s = std::atof(last.c_str());
            auto t0 = std::chrono::steady_clock::now();
            size_t n = 0;
            unsigned long long sink = 0;
            for (auto& q : queries) {
                json j = engine.search(q, K);
                drop_cache(engine);
                if (j.count("found")) sink += j["found"].get<uint64_t>();
                n++;
                double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (el > max_s) break;
            }
            double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("{\"queries\": %zu, \"seconds\": %.6f, \"qps\": %.6f, \"found_sum\": %llu}\n",
                        n, el, n / el, sink);
        } else {
            rc = 2;
        }
        drop_cache(engine);  // destructor saves caches; keep that trivial
    }
    // best-effort cleanup of the scratch directory
    std::remove("search_cache.json");
    std::remove("ai_overview_cache.json");
    std::remove("ai_summary_cache.json");
    if (chdir("/") == 0) rmdir(scratch);
    return rc;
}

// ns_tool — small CLI over the host facade.
//   ns_tool gen-index <index_dir> <n_segments> <docs_per_segment> [vocab=65536] [seed=1337] [--legacy]
//   ns_tool search <index_dir> <k> <query text ...>        (needs an MI355X; prints the /api/search JSON body)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

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
    std::fprintf(stderr, "usage: %s gen-index <dir> <n_segments> <docs_per_segment> [vocab] [seed] [--legacy]\n       %s search <dir> <k> <query...>\n", argv[0], argv[0]);
    return 2;
}

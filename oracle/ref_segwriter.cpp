// TEST INFRASTRUCTURE ONLY — drives the REAL reference's SegmentWriter (include/segment_writer.hpp, header-only,
// compiled where it lies under /root/reference by `make -C oracle ref`) so that tests can byte-compare this
// repo's loader / inversion against files the reference's own writer produced (SURVEY.md §4 tier T0).
// This file is ours; nothing of the reference is copied.
//
//   ref_segwriter <spec.tsv> <index_dir>
// spec: one document per line: cord_uid \t title \t json_relpath \t doc_len \t term:tf term:tf ...
// writes <index_dir>/segments/seg_000000/* through SegmentWriter::write_segment and <index_dir>/manifest.bin.
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "segment_writer.hpp"

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s <spec.tsv> <index_dir>\n", argv[0]); return 2; }
    std::ifstream in(argv[1]);
    if (!in) { std::perror("spec"); return 1; }
    SegmentWriter w;
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::vector<std::string> f;
        size_t pos = 0;
        for (int i = 0; i < 4; i++) {
            size_t t = line.find('\t', pos);
            if (t == std::string::npos) { std::fprintf(stderr, "bad spec line\n"); return 1; }
            f.push_back(line.substr(pos, t - pos));
            pos = t + 1;
        }
        DocMeta m{f[0], f[1], f[2], (uint32_t)std::stoul(f[3])};
        std::vector<std::pair<std::string, uint32_t>> tfs;
        std::istringstream rest(line.substr(pos));
        std::string tok;
        while (rest >> tok) {
            size_t c = tok.rfind(':');
            tfs.push_back({tok.substr(0, c), (uint32_t)std::stoul(tok.substr(c + 1))});
        }
        w.add_document(m, tfs);
    }
    fs::path idx = argv[2];
    w.write_segment(idx / "segments" / "seg_000000");
    std::ofstream man(idx / "manifest.bin", std::ios::binary);
    write_u32(man, 1);
    write_string(man, "seg_000000");
    return 0;
}

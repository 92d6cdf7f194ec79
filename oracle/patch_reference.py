#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (build container only: needs /root/reference).

Applies the reference-side patch of INTEGRATION.md section 1 to a SCRATCH copy of the two reference files it touches —
src/api_engine.cpp and include/api_engine.hpp — so that the drop-in claim is something a compiler has checked: the
patched cord19::Engine keeps its interface (reload(), search(query, k) -> json) and answers through
libnextsearch_hip.so instead of its own posting loop (src/api_engine.cpp:426-504).

    python3 oracle/patch_reference.py <reference root> <scratch dir>

Writes <scratch dir>/api_engine.cpp and <scratch dir>/api_engine.hpp.  The scratch dir lies under oracle/_ref/, which is
git-ignored: nothing of the reference's text enters the repository — this script holds only the lines the patch ADDS and
the short anchors it looks for.  oracle/Makefile (target `ref_gpu`) compiles the result, with every other reference
translation unit taken where it lies, into oracle/_ref/ref_driver_gpu (same CLI as ref_driver).
"""
import os
import sys

HPP_INCLUDE_ANCHOR = '#include "semantic_embedding.hpp"'
HPP_MEMBER_ANCHOR = "    std::mutex mtx;"
HPP_MEMBERS = '''
    // ---- MI355X drop-in (INTEGRATION.md section 1) ----
    ns_ctx* gpu = nullptr;                             // one device context (not re-entrant; Engine::mtx already serialises)
    std::vector<ns_seg*> gpu_segs;                     // one per loaded segment, same index as `segments`
    std::vector<std::vector<uint64_t>> barrel_base;    // byte offset of every inverted_bNNN.bin inside the uploaded payload
'''

CPP_INCLUDE_ANCHOR = '#include "textutil.hpp"'
CPP_INCLUDES = '''
#include <fcntl.h>
#include <stdexcept>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
'''

RELOAD_ANCHOR = "    segments = std::move(loaded);"
RELOAD_PATCH = '''    // ---- MI355X drop-in: a FRESH device context is filled from `loaded` and swapped in together with it ----
    {
        ns_ctx* fresh = nullptr;
        if (ns_ctx_create(/*device*/ 0, &fresh) != NS_OK) { std::cerr << ns_last_error(nullptr) << "\\n"; return false; }
        std::vector<ns_seg*> fresh_segs(loaded.size(), nullptr);
        std::vector<std::vector<uint64_t>> fresh_base(loaded.size());
        bool ok = true;
        for (uint32_t i = 0; i < (uint32_t)loaded.size() && ok; i++) {
            Segment& sg = loaded[i];
            std::vector<uint32_t> doc_len(sg.N, 0u);
            for (uint32_t d = 0; d < sg.N && d < (uint32_t)sg.docs.size(); d++) doc_len[d] = sg.docs[d].doc_len;
            std::vector<fs::path> files;                                  // the loader only OPENED them (api_segment.cpp:62-64,:74-78)
            if (sg.use_barrels) for (uint32_t b = 0; b < sg.barrel_params.barrel_count; b++) files.push_back(inv_barrel_path(sg.dir, b));
            else files.push_back(sg.dir / "inverted.bin");
            std::vector<uint64_t> sizes;
            uint64_t total = 0;
            for (auto& f : files) {
                fresh_base[i].push_back(total);
                std::error_code ec;
                const uint64_t n = (uint64_t)fs::file_size(f, ec) & ~7ull;
                if (ec) { ok = false; break; }
                sizes.push_back(n);
                total += n;
            }
            if (!ok) break;
            if (ns_segment_upload_begin(fresh, i, sg.N, sg.avgdl, doc_len.data(), total, &fresh_segs[i]) != NS_OK) { ok = false; break; }
            for (size_t f = 0; f < files.size() && ok; f++) {             // the payload never exists in host memory as a whole
                if (sizes[f] == 0) continue;
                const int fd = ::open(files[f].c_str(), O_RDONLY);
                if (fd < 0) { ok = false; break; }
                void* m = ::mmap(nullptr, sizes[f], PROT_READ, MAP_PRIVATE, fd, 0);
                ::close(fd);
                if (m == MAP_FAILED) { ok = false; break; }
                const int rc = ns_segment_upload_append(fresh, fresh_segs[i], m, sizes[f]);
                ::munmap(m, sizes[f]);
                if (rc != NS_OK) ok = false;
            }
            if (ok && ns_segment_upload_end(fresh, fresh_segs[i]) != NS_OK) ok = false;
        }
        if (!ok) {
            std::cerr << "device upload failed: " << ns_last_error(fresh) << "\\n";
            ns_ctx_destroy(fresh);                                        // nothing of the running engine was touched
            return false;
        }
        if (gpu) ns_ctx_destroy(gpu);                                     // releases the old segments
        gpu = fresh;
        gpu_segs = std::move(fresh_segs);
        barrel_base = std::move(fresh_base);
    }
'''

SEARCH_BEGIN_ANCHOR = "    // Define a hit record to keep (score, segment, doc)"
SEARCH_END_ANCHOR = "    std::reverse(hits.begin(), hits.end());"
SEARCH_PATCH = '''    // ---- MI355X drop-in: :426-504 (heap, per-segment maps, posting loop) become one call ----
    struct Hit {
        float s;
        uint32_t segId;
        uint32_t docId;
    };
    (void)k1; (void)b;                                                    // the BM25 constants live in the kernels now
    std::vector<ns_term_ref> refs;
    for (uint32_t segId = 0; segId < (uint32_t)segments.size(); segId++)
        for (const auto& tw : qterms_w) {
            auto it = segments[segId].lex.find(tw.first);
            if (it == segments[segId].lex.end() || it->second.df == 0) continue;
            const LexEntry& e = it->second;
            ns_term_ref r;
            r.seg_id = segId;
            r.count = e.count;
            r.byte_off = (segments[segId].use_barrels ? barrel_base[segId][e.barrelId] : 0) + e.offset;
            r.idf = bm25_idf(segments[segId].N, e.df);                    // stays on the host (glibc logf)
            r.qweight = tw.second;
            refs.push_back(r);
        }
    ns_query_desc qd{0, (uint32_t)refs.size()};
    std::vector<ns_hit> gpu_hits((size_t)K);
    uint32_t nhits = 0;
    uint64_t total_found = 0;
    if (ns_search_batch(gpu, &qd, refs.data(), 1, (uint32_t)K, gpu_hits.data(), &nhits, &total_found, NS_FLAG_OR) != NS_OK)
        throw std::runtime_error(ns_last_error(gpu));                     // -> HTTP 500 like any other exception (api_server.cpp:76-84)
    std::vector<Hit> hits;
    for (uint32_t i = 0; i < nhits; i++) hits.push_back(Hit{gpu_hits[i].score, gpu_hits[i].seg_id, gpu_hits[i].doc_id});
'''


def one(lines, anchor, what):
    idx = [i for i, ln in enumerate(lines) if ln.rstrip("\n") == anchor]
    if len(idx) != 1:
        sys.exit(f"patch_reference: anchor for {what} found {len(idx)} times (expected once): {anchor!r}")
    return idx[0]


def main():
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(ref, "include", "api_engine.hpp")) as f:
        hpp = f.readlines()
    i = one(hpp, HPP_INCLUDE_ANCHOR, "the header's includes")
    hpp.insert(i + 1, '#include "nextsearch_hip.h"\n')
    i = one(hpp, HPP_MEMBER_ANCHOR, "Engine's members")
    hpp.insert(i + 1, HPP_MEMBERS)
    with open(os.path.join(out, "api_engine.hpp"), "w") as f:
        f.writelines(hpp)

    with open(os.path.join(ref, "src", "api_engine.cpp")) as f:
        cpp = f.readlines()
    i = one(cpp, CPP_INCLUDE_ANCHOR, "the source's includes")
    cpp.insert(i + 1, CPP_INCLUDES)
    i = one(cpp, RELOAD_ANCHOR, "reload()'s commit line")
    cpp.insert(i, RELOAD_PATCH)
    a = one(cpp, SEARCH_BEGIN_ANCHOR, "the start of the scoring section")
    b = one(cpp, SEARCH_END_ANCHOR, "the end of the scoring section")
    if not a < b:
        sys.exit("patch_reference: scoring section anchors out of order")
    cpp[a:b + 1] = [SEARCH_PATCH]
    with open(os.path.join(out, "api_engine.cpp"), "w") as f:
        f.writelines(cpp)
    print(f"patched api_engine.cpp / api_engine.hpp written to {out}")


if __name__ == "__main__":
    main()

// On-disk NextSearch segment format: byte-compatible reader and writer (host side, no GPU).
//
// Format authority (reference, read as text; nothing copied):
//   manifest.bin                 src/api_segment.cpp:14-35          u32 n; n x string
//   segments/<seg>/stats.bin     src/api_segment.cpp:110-115        u32 N; f32 avgdl
//   segments/<seg>/docs.bin      src/api_segment.cpp:118-131        u32 n; n x {string uid,string title,string path,u32 doc_len}
//   segments/<seg>/barrels.bin   include/barrels.hpp:26-39          u32 barrel_count; u32 terms_per_barrel
//   .../lexicon_bNNN.bin         src/lexicon.cpp:116-120,131-147    u32 tcount; tcount x {string term,u32 termId,u32 df,u64 offset,u32 count}
//   .../inverted_bNNN.bin        src/lexicon.cpp:122-125            count x {u32 docId,u32 tf}, docId ascending
//   legacy lexicon.bin/inverted.bin  src/api_segment.cpp:45-67      same records, one file, no barrels
//   string = u32 len + bytes     include/indexio.hpp:18-29          little-endian, no padding
//
// Unlike the reference loader (which keeps 64 open ifstreams and seeks per term), this loader reads
// every inverted file once and flattens the barrels into ONE contiguous posting buffer plus a
// barrel_base[] table, because that buffer is what gets staged to HBM (include/nextsearch_hip.h).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace nsx {

namespace fs = std::filesystem;

static constexpr uint32_t kBarrelCount = 64;  // include/barrels.hpp:12

struct LexEntry {          // include/api_types.hpp:23-29
    uint32_t termId = 0;
    uint32_t df = 0;
    uint64_t offset = 0;   // byte offset inside its inverted file
    uint32_t count = 0;    // postings in the list (== df as written by the indexer, but a separate field)
    uint32_t barrelId = 0; // 0 for legacy segments
};

struct SegmentData {
    std::string name;
    fs::path dir;
    uint32_t N = 0;
    float avgdl = 0.0f;
    std::vector<uint32_t> doc_len;
    std::vector<std::string> cord_uid;
    std::unordered_map<std::string, LexEntry> lex;
    bool use_barrels = false;
    uint32_t barrel_count = 0;
    uint32_t terms_per_barrel = 0;
    std::vector<uint64_t> barrel_base;   // byte offset of each inverted file inside `postings`
    std::vector<uint8_t> postings;       // all inverted files back to back
    // absolute byte offset of a term's list inside `postings`
    uint64_t list_byte_offset(const LexEntry& e) const {
        return (use_barrels ? barrel_base[e.barrelId] : 0) + e.offset;
    }
};

// ---- little-endian file reader over a whole-file buffer -------------------------------------
class FileBytes {
public:
    bool load(const fs::path& p) {
        std::FILE* f = std::fopen(p.c_str(), "rb");
        if (!f) return false;
        std::fseek(f, 0, SEEK_END);
        long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        buf_.resize(n > 0 ? (size_t)n : 0);
        size_t got = buf_.empty() ? 0 : std::fread(buf_.data(), 1, buf_.size(), f);
        std::fclose(f);
        pos_ = 0;
        return got == buf_.size();
    }
    // Reads past the end yield zeros, as a failed ifstream::read leaves the reference's
    // uninitialised locals unspecified; we make that case deterministic instead.
    uint32_t u32() { uint32_t v = 0; take(&v, 4); return v; }
    uint64_t u64() { uint64_t v = 0; take(&v, 8); return v; }
    float f32() { float v = 0; take(&v, 4); return v; }
    std::string str() {
        uint32_t n = u32();
        if ((uint64_t)pos_ + n > buf_.size()) { pos_ = buf_.size(); return std::string(); }
        std::string s((const char*)buf_.data() + pos_, n);
        pos_ += n;
        return s;
    }
    void skip_str() { uint32_t n = u32(); pos_ = std::min(buf_.size(), pos_ + (size_t)n); }
    size_t size() const { return buf_.size(); }
    std::vector<uint8_t>& bytes() { return buf_; }
private:
    void take(void* dst, size_t n) {
        if (pos_ + n <= buf_.size()) { std::memcpy(dst, buf_.data() + pos_, n); pos_ += n; }
        else pos_ = buf_.size();
    }
    std::vector<uint8_t> buf_;
    size_t pos_ = 0;
};

inline std::string barrel_suffix(uint32_t b) {  // include/barrels.hpp:50-54 ("%03u")
    char buf[16];
    std::snprintf(buf, sizeof(buf), "%03u", b);
    return buf;
}
inline fs::path inv_barrel_path(const fs::path& d, uint32_t b) { return d / ("inverted_b" + barrel_suffix(b) + ".bin"); }
inline fs::path lex_barrel_path(const fs::path& d, uint32_t b) { return d / ("lexicon_b" + barrel_suffix(b) + ".bin"); }
inline bool has_barrels(const fs::path& d) {    // include/barrels.hpp:67-71
    return fs::exists(d / "barrels.bin") && fs::exists(inv_barrel_path(d, 0)) && fs::exists(lex_barrel_path(d, 0));
}
inline std::string seg_name(uint32_t id) {      // src/api_segment.cpp:38-42
    char buf[32];
    std::snprintf(buf, sizeof(buf), "seg_%06u", id);
    return buf;
}

inline std::vector<std::string> load_manifest(const fs::path& p) {
    std::vector<std::string> segs;
    FileBytes in;
    if (!fs::exists(p) || !in.load(p)) return segs;
    uint32_t n = in.u32();
    segs.reserve(n);
    for (uint32_t i = 0; i < n; i++) segs.push_back(in.str());
    return segs;
}

inline void read_lexicon_records(FileBytes& in, uint32_t barrel, std::unordered_map<std::string, LexEntry>& lex) {
    uint32_t tcount = in.u32();
    for (uint32_t i = 0; i < tcount; i++) {
        std::string term = in.str();
        LexEntry e;
        e.termId = in.u32();
        e.df = in.u32();
        e.offset = in.u64();
        e.count = in.u32();
        e.barrelId = barrel;
        lex.emplace(std::move(term), e);   // first occurrence wins, as unordered_map::emplace does in the reference
    }
}

// Mirrors load_segment (src/api_segment.cpp:105-136) but slurps the posting payload.
inline bool load_segment(const fs::path& segdir, SegmentData& s) {
    s = SegmentData{};
    s.dir = segdir;
    s.name = segdir.filename().string();
    {
        FileBytes in;
        if (!in.load(segdir / "stats.bin")) return false;
        s.N = in.u32();
        s.avgdl = in.f32();
    }
    {
        FileBytes in;
        if (!in.load(segdir / "docs.bin")) return false;
        uint32_t n = in.u32();
        s.doc_len.resize(n);
        s.cord_uid.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            s.cord_uid[i] = in.str();
            in.skip_str();
            in.skip_str();
            s.doc_len[i] = in.u32();
        }
    }
    if (has_barrels(segdir)) {
        s.use_barrels = true;
        FileBytes bm;
        if (!bm.load(segdir / "barrels.bin")) return false;
        s.barrel_count = bm.u32();
        s.terms_per_barrel = bm.u32();
        s.barrel_base.assign(s.barrel_count, 0);
        for (uint32_t b = 0; b < s.barrel_count; b++) {
            FileBytes inv;
            if (!inv.load(inv_barrel_path(segdir, b))) return false;
            s.barrel_base[b] = s.postings.size();
            s.postings.insert(s.postings.end(), inv.bytes().begin(), inv.bytes().end());
        }
        for (uint32_t b = 0; b < s.barrel_count; b++) {
            FileBytes in;
            if (!in.load(lex_barrel_path(segdir, b))) return false;
            read_lexicon_records(in, b, s.lex);
        }
        return true;
    }
    // legacy single-file layout
    s.use_barrels = false;
    FileBytes in;
    if (!in.load(segdir / "lexicon.bin")) return false;
    read_lexicon_records(in, 0, s.lex);
    FileBytes inv;
    if (!inv.load(segdir / "inverted.bin")) return false;
    s.postings = std::move(inv.bytes());
    return true;
}

// ---- writer ---------------------------------------------------------------------------------
class FileOut {
public:
    explicit FileOut(const fs::path& p) : f_(std::fopen(p.c_str(), "wb")) {
        if (!f_) throw std::runtime_error("cannot open for write: " + p.string());
    }
    ~FileOut() { if (f_) std::fclose(f_); }
    FileOut(const FileOut&) = delete;
    FileOut& operator=(const FileOut&) = delete;
    void u32(uint32_t v) { std::fwrite(&v, 4, 1, f_); }
    void u64(uint64_t v) { std::fwrite(&v, 8, 1, f_); }
    void f32(float v) { std::fwrite(&v, 4, 1, f_); }
    void str(const std::string& s) { u32((uint32_t)s.size()); if (!s.empty()) std::fwrite(s.data(), 1, s.size(), f_); }
    void raw(const void* p, size_t n) { if (n) std::fwrite(p, 1, n, f_); }
    void patch_u32_at0(uint32_t v) { std::fflush(f_); long cur = std::ftell(f_); std::fseek(f_, 0, SEEK_SET); u32(v); std::fseek(f_, cur, SEEK_SET); }
private:
    std::FILE* f_;
};

inline void save_manifest(const fs::path& p, const std::vector<std::string>& segs) {
    FileOut out(p);
    out.u32((uint32_t)segs.size());
    for (auto& s : segs) out.str(s);
}

}  // namespace nsx

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
// Like the reference loader (which keeps 64 open ifstreams and seeks per term, src/api_segment.cpp:70-102) this one
// does not hold the posting payload in host memory: it records each inverted file's size, lays the files out back to
// back in ONE logical posting buffer (barrel_base[]), and for_each_inverted_file() maps them one at a time so that
// the caller can feed the device's pinned staging copy straight from the page cache (include/nextsearch_hip.h:
// ns_segment_upload_begin / _append / _end).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <filesystem>
#include <functional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
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
    // The posting payload = all inverted files back to back (never resident on the host as a whole):
    std::vector<fs::path> inv_files;     // inverted_b000.bin .. (or the one inverted.bin), in payload order
    std::vector<uint64_t> inv_bytes;     // size of each, rounded down to whole {u32,u32} pairs
    std::vector<uint64_t> barrel_base;   // byte offset of each inverted file inside the payload
    uint64_t postings_bytes = 0;         // total payload size
    // absolute byte offset of a term's list inside the payload
    uint64_t list_byte_offset(const LexEntry& e) const {
        return (use_barrels ? barrel_base[e.barrelId] : 0) + e.offset;
    }
};

// Read-only mapping of a whole file (the posting payload is consumed from the page cache, not from a private copy).
class MappedFile {
public:
    MappedFile() = default;
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    ~MappedFile() { close(); }
    bool open(const fs::path& p) {
        close();
        const int fd = ::open(p.c_str(), O_RDONLY | O_CLOEXEC);
        if (fd < 0) return false;
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); return false; }
        size_ = (size_t)st.st_size;
        if (size_) {
            void* m = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); size_ = 0; return false; }
            (void)::madvise(m, size_, MADV_SEQUENTIAL);
            base_ = m;
        }
        ::close(fd);   // the mapping keeps the file
        return true;
    }
    void close() {
        if (base_) ::munmap(base_, size_);
        base_ = nullptr; size_ = 0;
    }
    const uint8_t* data() const { return (const uint8_t*)base_; }
    size_t size() const { return size_; }
private:
    void* base_ = nullptr;
    size_t size_ = 0;
};

// Hands the payload to `sink` piece by piece, in payload order: each inverted file is mapped, passed on and unmapped.
// Returns false if a file cannot be mapped or has changed size since load_segment, or when the sink returns false.
inline bool for_each_inverted_file(const SegmentData& s, const std::function<bool(const uint8_t*, uint64_t)>& sink) {
    for (size_t i = 0; i < s.inv_files.size(); i++) {
        MappedFile m;
        if (!m.open(s.inv_files[i]) || (m.size() & ~(size_t)7) != s.inv_bytes[i]) return false;
        if (s.inv_bytes[i] && !sink(m.data(), s.inv_bytes[i])) return false;
    }
    return true;
}
// The whole payload in one host buffer — for tests and tools that look at raw postings; the engine never does this.
inline bool read_postings(const SegmentData& s, std::vector<uint8_t>& out) {
    out.clear();
    out.reserve(s.postings_bytes);
    return for_each_inverted_file(s, [&](const uint8_t* p, uint64_t n) { out.insert(out.end(), p, p + n); return true; });
}

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
    if ((uint64_t)n * 4 > in.size()) return segs;               // corrupt count
    segs.reserve(n);
    for (uint32_t i = 0; i < n; i++) segs.push_back(in.str());
    return segs;
}

inline void read_lexicon_records(FileBytes& in, uint32_t barrel, std::unordered_map<std::string, LexEntry>& lex) {
    uint32_t tcount = in.u32();
    if ((uint64_t)tcount * 24 > in.size()) tcount = (uint32_t)(in.size() / 24);   // a record takes >= 24 bytes; reads past the end yield nothing anyway
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

// Mirrors load_segment (src/api_segment.cpp:105-136): stats, docs, lexicon in memory; the inverted files stay on disk.
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
        if ((uint64_t)n * 16 > in.size()) return false;          // a document record takes >= 16 bytes: the count is corrupt
        s.doc_len.resize(n);
        s.cord_uid.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            s.cord_uid[i] = in.str();
            in.skip_str();
            in.skip_str();
            s.doc_len[i] = in.u32();
        }
    }
    auto add_inverted = [&](const fs::path& p) -> bool {
        std::error_code ec;
        const uint64_t n = (uint64_t)fs::file_size(p, ec);
        if (ec) return false;
        s.inv_files.push_back(p);
        s.inv_bytes.push_back(n & ~7ull);
        s.postings_bytes += n & ~7ull;
        return true;
    };
    if (has_barrels(segdir)) {
        s.use_barrels = true;
        FileBytes bm;
        if (!bm.load(segdir / "barrels.bin")) return false;
        s.barrel_count = bm.u32();
        s.terms_per_barrel = bm.u32();
        if (s.barrel_count > 4096) return false;                 // the reference writes 64 (include/barrels.hpp:12): a corrupt header
        s.barrel_base.assign(s.barrel_count, 0);
        for (uint32_t b = 0; b < s.barrel_count; b++) {
            s.barrel_base[b] = s.postings_bytes;
            if (!add_inverted(inv_barrel_path(segdir, b))) return false;
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
    return add_inverted(segdir / "inverted.bin");
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

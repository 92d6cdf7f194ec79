// Result decoration from <index>/metadata.csv (the step right after the hot path).
//
// The reference keeps one byte offset per cord_uid (load_metadata_uid_meta,
// src/api_metadata.cpp:109-185) and, PER HIT, opens the CSV, seeks to the row, parses it, seeks
// back and parses the header again (fetch_metadata, :188-249; src/api_engine.cpp:516-531).  Here
// the file is read ONCE at reload(): the header is parsed once, every physical line is parsed
// once, and the four decorated fields of the rows that belong to loaded documents are kept in one
// string arena, indexed per (segment, docId).  Decorating a hit is then two array reads.
//
// Semantics kept bit for bit (they are what makes the JSON equal):
//   * a row is one PHYSICAL line (std::getline): a quoted field with an embedded newline is split,
//     exactly as the reference splits it;
//   * csv splitting: '"' toggles the quoted state and is dropped, ',' splits outside quotes
//     (src/api_metadata.cpp:13-43) — no "" escape;
//   * header: the LAST column named cord_uid / url / publish_time / authors / title wins (:144-147, :223-229);
//   * rows with fewer columns than the cord_uid column, or an empty cord_uid, are skipped; the FIRST
//     row of a cord_uid wins (:156-176);
//   * author = first_author_et_al(authors) (:58-106); url is cut at its first ';'
//     (src/api_engine.cpp:525-527); a field is emitted only when non-empty (:523-531).
#pragma once

#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "index_format.hpp"

namespace nsx {

struct MetaFields {   // what one hit gets; empty = key absent from the JSON
    std::string title, url, publish_time, author;
};

// One CSV record cut into fields.  The fields' bytes live back to back in ONE buffer (quotes already gone); field i
// is the span [cut[i], cut[i + 1]).  Rules (src/api_metadata.cpp:13-43): a '"' flips the quoted state and vanishes,
// a ',' outside quotes ends a field, nothing is escaped; a record always has at least one field.
struct CsvRecord {
    std::string bytes;
    std::vector<uint32_t> cut;
    void parse(std::string_view line) {
        bytes.clear();
        cut.assign(1, 0u);
        bytes.reserve(line.size());
        unsigned quoted = 0;
        for (const char ch : line) {
            if (ch == '"') quoted ^= 1u;
            else if (ch == ',' && !quoted) cut.push_back((uint32_t)bytes.size());
            else bytes.push_back(ch);
        }
        cut.push_back((uint32_t)bytes.size());
    }
    size_t size() const { return cut.size() - 1; }
    std::string_view operator[](size_t i) const { return std::string_view(bytes).substr(cut[i], cut[i + 1] - cut[i]); }
    // field i, or nothing when the record is too short or the column does not exist (col < 0)
    std::string_view at(int col) const { return (col >= 0 && (size_t)col < size()) ? (*this)[(size_t)col] : std::string_view(); }
};

namespace detail {
// the "C" locale's isspace set (what the reference's std::isspace sees): ' ', \t \n \v \f \r
constexpr bool blank(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
inline std::string_view strip(std::string_view v) {
    while (!v.empty() && blank(v.front())) v.remove_prefix(1);
    while (!v.empty() && blank(v.back())) v.remove_suffix(1);
    return v;
}
}  // namespace detail

// The "author" field of a hit (src/api_metadata.cpp:58-106): the surname of the first author of CORD-19's
// `authors` column plus " et al.", or nothing.  Authors are ';'-separated; one author is "Surname, Given" or
// "Given ... Surname"; a leading "(...)" holds a romanised spelling that replaces the whole author when non-blank.
inline std::string first_author_et_al(std::string_view authors) {
    using detail::strip;
    std::string_view who = strip(authors);
    who = strip(who.substr(0, who.find(';')));                       // find() == npos keeps everything
    while (!who.empty() && (who.back() == ',' || detail::blank(who.back()))) who.remove_suffix(1);
    who = strip(who);
    if (who.empty()) return {};
    if (who.front() == '(') {
        const size_t rp = who.find(')');
        if (rp != std::string_view::npos && rp > 1) {
            const std::string_view roman = strip(who.substr(1, rp - 1));
            if (!roman.empty()) who = roman;
        }
    }
    std::string_view family;
    if (const size_t c = who.find(','); c != std::string_view::npos) {
        family = strip(who.substr(0, c));
    } else {
        const size_t gap = who.find_last_of(" \t");
        family = strip(gap == std::string_view::npos ? who : who.substr(gap + 1));
    }
    if (family.empty()) return {};
    std::string label(family);
    label += " et al.";
    return label;
}

class MetadataTable {
public:
    // per segment, per docId: index into rows_ (0 = none)
    std::vector<std::vector<uint32_t>> doc_row;
    size_t rows_loaded = 0, rows_in_file = 0;

    void clear() { doc_row.clear(); rows_.assign(1, MetaFields{}); rows_loaded = rows_in_file = 0; }

    // Reads <index>/metadata.csv once.  The file is cut at line ends into one slice per host thread; every slice
    // parses its physical lines on its own (the header's column numbers and the uid -> documents map are read-only),
    // and the slices' rows are then merged in FILE ORDER, which is what "the first row of a cord_uid wins" refers to.
    bool load(const fs::path& csv, const std::vector<SegmentData>& segments, unsigned max_threads = 0) {
        clear();
        doc_row.resize(segments.size());
        for (size_t s = 0; s < segments.size(); s++) doc_row[s].assign(segments[s].cord_uid.size(), 0u);
        FileBytes f;
        if (!f.load(csv)) return false;
        const std::string_view text((const char*)f.bytes().data(), f.size());
        if (text.empty()) return false;                      // std::getline of the header fails
        // header line
        size_t body = text.find('\n');
        const std::string_view head = text.substr(0, body);
        body = (body == std::string_view::npos) ? text.size() : body + 1;
        struct Cols { int uid = -1, url = -1, pub = -1, auth = -1, title = -1; } col;
        {
            CsvRecord h;
            h.parse(head);
            for (size_t i = 0; i < h.size(); i++) {          // the LAST column of a name wins
                const std::string_view name = h[i];
                if (name == "cord_uid") col.uid = (int)i;
                else if (name == "url") col.url = (int)i;
                else if (name == "publish_time") col.pub = (int)i;
                else if (name == "authors") col.auth = (int)i;
                else if (name == "title") col.title = (int)i;
            }
        }
        if (col.uid < 0) return false;
        // cord_uid -> every (segment, docId) that carries it
        std::unordered_map<std::string_view, std::vector<std::pair<uint32_t, uint32_t>>> wanted;
        {
            size_t n_docs = 0;
            for (const auto& sg : segments) n_docs += sg.cord_uid.size();
            wanted.reserve(n_docs);
            for (uint32_t s = 0; s < segments.size(); s++)
                for (uint32_t d = 0; d < segments[s].cord_uid.size(); d++) wanted[segments[s].cord_uid[d]].push_back({s, d});
        }
        // slices [cutp[i], cutp[i+1]) of the body, each starting right after a '\n'
        unsigned nt = max_threads ? max_threads : std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
        size_t min_slice = 4u << 20;                         // below ~4 MB per thread the spawn costs more than it saves
        if (const char* e = std::getenv("NS_META_SLICE_BYTES")) min_slice = std::max<size_t>(1, std::strtoull(e, nullptr, 10));   // tests: many slices of a small file
        nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, (text.size() - body) / min_slice));
        std::vector<size_t> cutp(nt + 1, text.size());
        cutp[0] = body;
        for (unsigned i = 1; i < nt; i++) {
            size_t at = body + (text.size() - body) / nt * i;
            at = std::max(at, cutp[i - 1]);
            const size_t nl = text.find('\n', at);
            cutp[i] = (nl == std::string_view::npos) ? text.size() : nl + 1;
        }
        struct Row { std::string_view uid; MetaFields m; };
        struct Slice { std::vector<Row> rows; size_t lines = 0; };
        std::vector<Slice> slices(nt);
        auto parse_slice = [&](unsigned i) {
            Slice& out = slices[i];
            CsvRecord rec;
            size_t pos = cutp[i];
            const size_t stop = cutp[i + 1];
            while (pos < stop) {                             // std::getline: a last line without '\n' still counts
                size_t e = text.find('\n', pos);
                if (e == std::string_view::npos || e > stop) e = stop;
                const std::string_view line = text.substr(pos, e - pos);
                pos = e + 1;
                out.lines++;
                rec.parse(line);
                const std::string_view uid = rec.at(col.uid);
                if (uid.empty()) continue;                   // short row, or an empty cord_uid
                const auto w = wanted.find(uid);
                if (w == wanted.end()) continue;             // not a loaded document: nothing to keep
                Row r;
                r.uid = w->first;                            // the segment's own copy outlives `rec`
                r.m.title = std::string(rec.at(col.title));
                const std::string_view url = rec.at(col.url);
                r.m.url = std::string(url.substr(0, url.find(';')));   // src/api_engine.cpp:525-527
                r.m.publish_time = std::string(rec.at(col.pub));
                r.m.author = first_author_et_al(rec.at(col.auth));
                out.rows.push_back(std::move(r));
            }
        };
        if (nt == 1) {
            parse_slice(0);
        } else {
            std::vector<std::thread> pool;
            for (unsigned i = 0; i < nt; i++) pool.emplace_back(parse_slice, i);
            for (auto& t : pool) t.join();
        }
        std::unordered_set<std::string_view> taken;          // a cord_uid's first row (in file order) wins
        for (Slice& sl : slices) {
            rows_in_file += sl.lines;
            for (Row& r : sl.rows) {
                if (!taken.insert(r.uid).second) continue;
                rows_.push_back(std::move(r.m));
                const uint32_t idx = (uint32_t)rows_.size() - 1;
                for (const auto& sd : wanted.find(r.uid)->second) doc_row[sd.first][sd.second] = idx;
                rows_loaded++;
            }
        }
        return true;
    }

    // nullptr: the document has no metadata row (the reference's uid_to_meta.find() == end())
    const MetaFields* get(uint32_t seg, uint32_t doc) const {
        if (seg >= doc_row.size() || doc >= doc_row[seg].size()) return nullptr;
        const uint32_t i = doc_row[seg][doc];
        return i ? &rows_[i] : nullptr;
    }

private:
    std::vector<MetaFields> rows_{1};
};

}  // namespace nsx

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
#include <string>
#include <unordered_map>
#include <vector>

#include "index_format.hpp"

namespace nsx {

struct MetaFields {   // what one hit gets; empty = key absent from the JSON
    std::string title, url, publish_time, author;
};

inline void csv_split(const char* p, size_t n, std::vector<std::string>& out) {
    out.clear();
    std::string cur;
    bool inq = false;
    for (size_t i = 0; i < n; i++) {
        const char c = p[i];
        if (c == '"') { inq = !inq; continue; }
        if (!inq && c == ',') { out.push_back(cur); cur.clear(); continue; }
        cur.push_back(c);
    }
    out.push_back(cur);
}

inline std::string trim_ws(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) a++;
    while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

// "Surname et al." from the CORD-19 authors column ("Surname, Given; Surname2, Given2" or "Given Surname")
inline std::string first_author_et_al(const std::string& raw) {
    std::string s = trim_ws(raw);
    if (s.empty()) return "";
    const size_t semi = s.find(';');
    std::string first = trim_ws(semi == std::string::npos ? s : s.substr(0, semi));
    while (!first.empty() && (first.back() == ',' || std::isspace((unsigned char)first.back()))) first.pop_back();
    first = trim_ws(first);
    if (first.empty()) return "";
    if (first.front() == '(') {   // romanised name in parentheses
        const size_t close = first.find(')');
        if (close != std::string::npos && close > 1) {
            const std::string inside = trim_ws(first.substr(1, close - 1));
            if (!inside.empty()) first = inside;
        }
    }
    std::string surname;
    const size_t comma = first.find(',');
    if (comma != std::string::npos) {
        surname = trim_ws(first.substr(0, comma));
    } else {
        const std::string tmp = trim_ws(first);
        const size_t sp = tmp.find_last_of(" \t");
        surname = (sp == std::string::npos) ? tmp : trim_ws(tmp.substr(sp + 1));
    }
    surname = trim_ws(surname);
    if (surname.empty()) return "";
    return surname + " et al.";
}

class MetadataTable {
public:
    // per segment, per docId: index into rows_ (0 = none)
    std::vector<std::vector<uint32_t>> doc_row;
    size_t rows_loaded = 0, rows_in_file = 0;

    void clear() { doc_row.clear(); rows_.assign(1, MetaFields{}); rows_loaded = rows_in_file = 0; }

    // `wanted`: cord_uid -> every (segment, docId) that carries it
    bool load(const fs::path& csv, const std::vector<SegmentData>& segments) {
        clear();
        doc_row.resize(segments.size());
        for (size_t s = 0; s < segments.size(); s++) doc_row[s].assign(segments[s].cord_uid.size(), 0u);
        FileBytes f;
        if (!f.load(csv)) return false;
        const char* p = (const char*)f.bytes().data();
        const size_t n = f.size();
        std::unordered_map<std::string, std::vector<std::pair<uint32_t, uint32_t>>> wanted;
        wanted.reserve(1024);
        for (uint32_t s = 0; s < segments.size(); s++)
            for (uint32_t d = 0; d < segments[s].cord_uid.size(); d++) wanted[segments[s].cord_uid[d]].push_back({s, d});
        size_t pos = 0;
        auto next_line = [&](const char*& lp, size_t& ln) -> bool {   // std::getline semantics
            if (pos >= n) return false;
            size_t e = pos;
            while (e < n && p[e] != '\n') e++;
            lp = p + pos; ln = e - pos;
            pos = (e < n) ? e + 1 : n;
            return true;
        };
        const char* lp; size_t ln;
        if (!next_line(lp, ln)) return false;
        std::vector<std::string> cols, r;
        csv_split(lp, ln, cols);
        int uid_i = -1, url_i = -1, pub_i = -1, auth_i = -1, title_i = -1;
        for (int i = 0; i < (int)cols.size(); i++) {
            if (cols[i] == "cord_uid") uid_i = i;
            if (cols[i] == "url") url_i = i;
            if (cols[i] == "publish_time") pub_i = i;
            if (cols[i] == "authors") auth_i = i;
            if (cols[i] == "title") title_i = i;
        }
        if (uid_i < 0) return false;
        std::unordered_map<std::string, bool> seen;   // first occurrence of a cord_uid wins, wanted or not
        while (next_line(lp, ln)) {
            rows_in_file++;
            csv_split(lp, ln, r);
            if ((int)r.size() <= uid_i) continue;
            const std::string& uid = r[uid_i];
            if (uid.empty()) continue;
            auto w = wanted.find(uid);
            if (w == wanted.end()) continue;           // not a loaded document: nothing to keep
            if (!seen.emplace(uid, true).second) continue;
            MetaFields m;
            if (title_i >= 0 && (int)r.size() > title_i) m.title = r[title_i];
            if (url_i >= 0 && (int)r.size() > url_i) {
                m.url = r[url_i];
                const size_t semi = m.url.find(';');
                if (semi != std::string::npos) m.url.resize(semi);
            }
            if (pub_i >= 0 && (int)r.size() > pub_i) m.publish_time = r[pub_i];
            if (auth_i >= 0 && (int)r.size() > auth_i) m.author = first_author_et_al(r[auth_i]);
            rows_.push_back(std::move(m));
            const uint32_t idx = (uint32_t)rows_.size() - 1;
            for (auto& sd : w->second) doc_row[sd.first][sd.second] = idx;
            rows_loaded++;
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

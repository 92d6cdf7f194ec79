// Query text -> scored terms, restating the reference's behaviour byte for byte (it decides which
// terms reach the kernels):
//   tokenize     include/textutil.hpp:13-28   maximal runs of isalnum() bytes, each tolower()'d;
//                                             in the "C" locale that is exactly [0-9A-Za-z], bytes
//                                             >= 0x80 split tokens
//   is_stopword  include/textutil.hpp:31-37   24-word list
//   base terms   src/api_engine.cpp:391-397   drop size() < 2 and stopwords; order and duplicates kept
#pragma once

#include <string>
#include <vector>

namespace nextsearch {

inline bool is_alnum_ascii(unsigned char c) {
    return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
}

inline std::vector<std::string> tokenize(const std::string& text) {
    std::vector<std::string> out;
    std::string cur;
    for (unsigned char c : text) {
        if (is_alnum_ascii(c)) {
            cur.push_back((c >= 'A' && c <= 'Z') ? (char)(c - 'A' + 'a') : (char)c);
        } else if (!cur.empty()) {
            out.push_back(cur);
            cur.clear();
        }
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

inline bool is_stopword(const std::string& t) {
    static const char* const kStop[] = {"the", "a",  "an",  "and",  "or", "of",   "to", "in",   "for",  "on",   "with", "by",
                                        "as",  "is", "are", "was",  "were", "be", "been", "it", "this", "that", "from", "at"};
    for (const char* s : kStop)
        if (t == s) return true;
    return false;
}

inline std::vector<std::string> base_terms(const std::string& query) {
    std::vector<std::string> out;
    for (auto& t : tokenize(query)) {
        if (t.size() < 2) continue;
        if (is_stopword(t)) continue;
        out.push_back(t);
    }
    return out;
}

}  // namespace nextsearch

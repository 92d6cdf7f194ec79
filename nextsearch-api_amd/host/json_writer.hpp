// Minimal JSON text writer for the /api/search surface.  The reference builds an nlohmann::json
// object (keys therefore serialise ALPHABETICALLY) and the HTTP layer prints it with dump(2)
// (src/api_engine.cpp:400-404,505-536; src/api_server.cpp:177).  This writer reproduces that text
// layout for the fields the hot path produces; it is not a general JSON library.
#pragma once

#include <charconv>
#include <cstdint>
#include <cstdio>
#include <string>

namespace nextsearch {

inline void json_escape(std::string& out, const std::string& s) {
    out.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default:
                if (c < 0x20) {
                    char buf[8];
                    std::snprintf(buf, sizeof(buf), "\\u%04x", c);
                    out += buf;
                } else {
                    out.push_back((char)c);
                }
        }
    }
    out.push_back('"');
}

// A float stored in a json number is widened to double and printed as the shortest string that
// round-trips the DOUBLE (so 2.2f prints as 2.200000047683716); integral values get ".0".
inline void json_number_from_float(std::string& out, float f) {
    double d = (double)f;
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), d);
    std::string s(buf, r.ptr);
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";   // 'n' guards inf/nan spellings
    out += s;
}

}  // namespace nextsearch

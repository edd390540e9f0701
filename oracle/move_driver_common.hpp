// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY.
// Input preparation shared by ref_driver_rlc.cpp (the reference's MoveLFReprBP) and oracle_driver.cpp (the
// restatement): text -> naive suffix array -> BWT codes and cumulative character counts.  Ours; no reference code.
// ============================================================================
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

namespace movedrv {

inline int code(char c) { return c == '$' ? 0 : c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 3 : 4; }

struct Prepared {
    std::string text; // with the final '$'
    std::vector<uint32_t> sa;
    std::vector<uint8_t> bwt;
    uint32_t cum[5];
};

inline Prepared prepare(std::string t, bool reversed = false) {
    Prepared p;
    if (t.empty() || t.back() != '$') t.push_back('$');
    std::string s = t;
    if (reversed) std::reverse(s.begin(), s.end()); // buildindex.cpp:749-758: the '$' comes first in the reversed text
    const uint32_t n = (uint32_t)s.size();
    p.sa.resize(n);
    for (uint32_t i = 0; i < n; i++) p.sa[i] = i;
    std::sort(p.sa.begin(), p.sa.end(), [&](uint32_t a, uint32_t b) { return s.compare(a, std::string::npos, s, b, std::string::npos) < 0; });
    p.bwt.resize(n);
    for (uint32_t i = 0; i < n; i++) // buildindex.cpp:706-712 (generateBWT), :575-585 (createRevBWT)
        p.bwt[i] = (uint8_t)code(reversed ? (p.sa[i] > 0 ? t[n - p.sa[i]] : t.front()) : (p.sa[i] > 0 ? t[p.sa[i] - 1] : t.back()));
    uint32_t cnt[5] = {0, 0, 0, 0, 0};
    for (char c : t) cnt[code(c)]++;
    uint32_t tot = 0;
    for (int c = 0; c < 5; c++) {
        p.cum[c] = tot;
        tot += cnt[c];
    }
    p.text = t;
    return p;
}

inline void hexBytes(std::ostream& os, const std::vector<uint8_t>& b) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (uint8_t v : b) {
        s.push_back(d[v >> 4]);
        s.push_back(d[v & 15]);
    }
    os << s;
}

} // namespace movedrv

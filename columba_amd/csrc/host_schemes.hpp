// Host side of the strategy: Search metadata, search-scheme tables, scheme-directory loaders.
//
// Mirrors (reference, src/):
//   Search::makeSearch                         search.h:116-194
//   SearchScheme sanity checks / critical part search.h:525-588
//   SearchScheme::readScheme                   search.h:599-711
//   KucherovKPlus1 tables                      searchstrategy.h:2829-2913
//   PigeonHoleSearchStrategy tables            searchstrategy.h:3221-3274
//   MultipleSchemesStrategy::readSchemes       searchstrategy.h:2624-2660
//   CustomSearchStrategy::getSearchSchemeFromFolder   searchstrategy.cpp:1990 ff.
//   base-class partition defaults              searchstrategy.h:245, :283, :1825
#pragma once
#include "dev_search.hpp"

#include <algorithm>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace cmb {

struct HostSearch {
    std::vector<uint32_t> pi, L, U;
    uint32_t sIdx = 0;
    // "heavier U-string first" ordering used to find the critical search (search.h:422-441)
    bool before(const HostSearch& o) const {
        for (size_t i = 0; i < U.size(); i++)
            if (U[i] != o.U[i]) return U[i] > o.U[i];
        for (size_t i = 0; i < L.size(); i++)
            if (L[i] != o.L[i]) return L[i] < o.L[i];
        return sIdx < o.sIdx;
    }
};

template <int MP = MAXP_WIDE>
inline DevSearchT<MP> toDevSearch(const HostSearch& h) {
    const size_t n = h.pi.size();
    if (n != h.L.size() || n != h.U.size())
        throw std::runtime_error("Could not create search, the sizes of all vectors are not equal");
    if (n < 1 || n > (size_t)MP) throw std::runtime_error("unsupported number of parts in search");
    DevSearchT<MP> d{};
    d.n = (uint8_t)n;
    for (size_t i = 0; i < n; i++) {
        d.order[i] = (uint8_t)h.pi[i];
        d.L[i] = (uint8_t)h.L[i];
        d.U[i] = (uint8_t)h.U[i];
    }
    // directions: phase 0 copies phase 1 (search.h:131)
    d.dir[0] = (n > 1 && h.pi[1] > h.pi[0]) ? 0 : 1; // (one part, `-S naive`: the search is never run)
    for (size_t i = 1; i < n; i++) d.dir[i] = h.pi[i] > h.pi[i - 1] ? 0 : 1;
    d.dsw[0] = d.dsw[1] = 0;
    for (size_t i = 2; i < n; i++) d.dsw[i] = d.dir[i] != d.dir[i - 1];
    uint32_t lo = h.pi[0], hi = h.pi[0];
    d.low[0] = (uint8_t)lo;
    d.high[0] = (uint8_t)hi;
    for (size_t i = 1; i < n; i++) {
        if (h.pi[i] < lo) lo = h.pi[i];
        else hi = h.pi[i];
        d.low[i] = (uint8_t)lo;
        d.high[i] = (uint8_t)hi;
    }
    d.uniAll = h.pi[0] == n - 1;
    d.uniIdx = (uint8_t)n;
    if (h.pi[n - 1] == 0 && !d.uniAll) {
        for (size_t i = 0; i < n; i++)
            if (h.pi[i] == n - 1) {
                d.uniIdx = (uint8_t)(i + 1);
                break;
            }
    } else if (h.pi[n - 1] == 0 && d.uniAll) {
        d.uniIdx = 0;
    }
    return d;
}

struct HostScheme {
    std::vector<HostSearch> searches;
    uint32_t k = 0;
    uint32_t critical = 0;
    uint32_t numParts() const { return (uint32_t)searches.front().pi.size(); }

    void finalize() {
        if (searches.empty()) throw std::runtime_error("Empty scheme");
        const size_t P = searches.front().pi.size();
        for (const auto& s : searches) {
            if (s.pi.size() != P)
                throw std::runtime_error("Not all searches for distance " + std::to_string(k) +
                                         " have the same number of parts");
            if (*std::min_element(s.pi.begin(), s.pi.end()) != 0)
                throw std::runtime_error("Not all searches are zero based for distance " + std::to_string(k) + "!");
            uint32_t hi = s.pi[0], lo = s.pi[0];
            for (size_t i = 1; i < P; i++) {
                if (s.pi[i] == hi + 1) hi++;
                else if (s.pi[i] + 1 == lo) lo--;
                else
                    throw std::runtime_error("Connectivity property not satisfied for all searches with distance " +
                                             std::to_string(k) + "!");
            }
            bool okb = s.L[0] <= s.U[0];
            for (size_t i = 1; i < P && okb; i++)
                okb = !(s.L[i] > s.U[i] || s.L[i] < s.L[i - 1] || s.U[i] < s.U[i - 1]);
            if (!okb)
                throw std::runtime_error("Decreasing lower or upper bounds for a search for K  = " +
                                         std::to_string(k));
        }
        size_t best = 0;
        for (size_t i = 1; i < searches.size(); i++)
            if (searches[i].before(searches[best])) best = i;
        critical = searches[best].pi[0];
    }
};

// "{0,1,2}" -> vector (search.h:635-650)
inline std::vector<uint32_t> parseBraces(const std::string& tok) {
    if (tok.size() < 2) throw std::runtime_error(tok + " is not a valid vector for a search");
    std::stringstream ss(tok.substr(1, tok.size() - 2));
    std::string item;
    std::vector<uint32_t> v;
    while (std::getline(ss, item, ',')) v.push_back((uint32_t)std::stoull(item));
    return v;
}

inline HostScheme readSchemeFile(const std::string& path, uint32_t k) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Cannot open file: " + path);
    HostScheme sch;
    sch.k = k;
    std::string line;
    uint32_t idx = 0;
    while (std::getline(f, line)) {
        std::stringstream ss(line);
        std::vector<std::string> toks;
        std::string t;
        while (ss >> t) toks.push_back(t);
        if (line.empty()) continue; // (search.h:694)
        HostSearch s;
        try { // search.h:599-621 + Search::makeSearch :116-124, wrapped per line as in readScheme (:697-703)
            if (toks.size() != 3)
                throw std::runtime_error("A search should have 3 vectors: order, lower bound and upper bound!");
            s.pi = parseBraces(toks[0]);
            s.L = parseBraces(toks[1]);
            s.U = parseBraces(toks[2]);
            if (s.pi.size() != s.L.size() || s.pi.size() != s.U.size())
                throw std::runtime_error("Could not create search, the sizes of all vectors are not equal");
        } catch (const std::runtime_error& e) {
            throw std::runtime_error("Something went wrong with processing line: " + line + "\nin file: " + path + "\n" +
                                     e.what());
        }
        s.sIdx = idx++;
        sch.searches.push_back(s);
    }
    if (sch.searches.empty()) throw std::runtime_error("Empty scheme in: " + path);
    sch.finalize();
    return sch;
}

struct PartitionParams {
    std::vector<double> seeding, begins;
    std::vector<uint64_t> weights;
};

} // namespace cmb

// The strategy handle of the C-ABI
struct cmb_strategy {
    int metric = 1, partition = 2;
    uint32_t kmerCutOff = 20;
    std::map<uint32_t, std::vector<cmb::HostScheme>> schemes; // k -> alternatives (dynamic selection)
    std::map<uint32_t, cmb::PartitionParams> params;          // k -> overrides

    // partitioning parameters for distance k as the host sees them (any number of parts): base-class defaults
    // (searchstrategy.h:245, :283, :1825) overridden by the strategy's own
    cmb::PartitionParams partitionFor(uint32_t k) const {
        using namespace cmb;
        auto it = schemes.find(k);
        if (it == schemes.end() || it->second.empty())
            throw std::runtime_error("the search strategy does not support distance " + std::to_string(k));
        const int P = (int)it->second.front().numParts();
        PartitionParams out;
        for (int i = 1; i < P - 1; i++) out.seeding.push_back(i * (1.0 / (P - 1)));
        out.weights.assign(P, 1);
        out.weights[0] = 2;
        out.weights[P - 1] = 2;
        for (int i = 1; i < P; i++) out.begins.push_back(i * (1.0 / P));
        auto pp = params.find(k);
        if (pp != params.end()) {
            const auto& q = pp->second;
            if (!q.seeding.empty()) {
                if ((int)q.seeding.size() != P - 2) throw std::runtime_error("wrong number of seeding positions");
                out.seeding = q.seeding;
            }
            if (!q.weights.empty()) {
                if ((int)q.weights.size() != P) throw std::runtime_error("wrong number of weights");
                out.weights = q.weights;
            }
            if (!q.begins.empty()) {
                if ((int)q.begins.size() != P - 1) throw std::runtime_error("wrong number of static positions");
                out.begins = q.begins;
            }
        }
        return out;
    }

    // number of parts of the schemes for distance k (0: the strategy has none)
    uint32_t numPartsFor(uint32_t k) const {
        auto it = schemes.find(k);
        return it == schemes.end() || it->second.empty() ? 0u : it->second.front().numParts();
    }
    // flatten everything matchWithSearches needs for distance k into device tables of MP parts (MAXP: the common kernels,
    // MAXP_WIDE: the instances for schemes with more parts)
    template <int MP = cmb::MAXP>
    cmb::DevStrategyKT<MP> flatten(uint32_t k) const {
        using namespace cmb;
        typedef DevStrategyKT<MP> DevStrategyK;
        auto it = schemes.find(k);
        if (it == schemes.end() || it->second.empty())
            throw std::runtime_error("the search strategy does not support distance " + std::to_string(k));
        const auto& alts = it->second;
        if (alts.size() > (size_t)MAXSCH) throw std::runtime_error("too many alternative schemes");
        if (alts.front().numParts() > (uint32_t)MP)
            throw std::runtime_error("search schemes with more than " + std::to_string(MP) + " parts (" + std::to_string(k) +
                                     " errors) cannot be run on the device: its tables hold " + std::to_string(MP) + " parts");
        DevStrategyK d{};
        d.metric = (uint8_t)metric;
        d.partition = (uint8_t)partition;
        d.kmerCutOff = kmerCutOff;
        d.numParts = (uint8_t)alts.front().numParts();
        d.nSchemes = (uint8_t)alts.size();
        const int P = d.numParts;
        // base-class defaults (searchstrategy.h:245, :283, :1825)
        for (int i = 1; i < P - 1; i++) d.seeding[i - 1] = i * (1.0 / (P - 1));
        for (int i = 0; i < P; i++) d.weights[i] = 1;
        d.weights[0] = 2;
        d.weights[P - 1] = 2;
        for (int i = 1; i < P; i++) d.begins[i - 1] = i * (1.0 / P);
        auto pp = params.find(k);
        if (pp != params.end()) {
            const auto& q = pp->second;
            if (!q.seeding.empty()) {
                if ((int)q.seeding.size() != P - 2) throw std::runtime_error("wrong number of seeding positions");
                for (size_t i = 0; i < q.seeding.size(); i++) d.seeding[i] = q.seeding[i];
            }
            if (!q.weights.empty()) {
                if ((int)q.weights.size() != P) throw std::runtime_error("wrong number of weights");
                for (size_t i = 0; i < q.weights.size(); i++) d.weights[i] = q.weights[i];
            }
            if (!q.begins.empty()) {
                if ((int)q.begins.size() != P - 1) throw std::runtime_error("wrong number of static positions");
                for (size_t i = 0; i < q.begins.size(); i++) d.begins[i] = q.begins[i];
            }
        }
        for (size_t a = 0; a < alts.size(); a++) {
            const HostScheme& h = alts[a];
            if (h.numParts() != (uint32_t)P) throw std::runtime_error("Not all schemes have same number of parts");
            if (h.searches.size() > (size_t)MAXS) throw std::runtime_error("too many searches in scheme");
            d.sch[a].nSearches = (uint8_t)h.searches.size();
            d.sch[a].critical = (uint8_t)h.critical;
            for (size_t i = 0; i < h.searches.size(); i++) d.sch[a].s[i] = toDevSearch<MP>(h.searches[i]);
        }
        return d;
    }
};

namespace cmb {

// ---- built-in tables (written as "pi L U" digit strings) -------------------------------------
inline HostScheme schemeFromRows(uint32_t k, std::initializer_list<const char*> rows) {
    HostScheme sch;
    sch.k = k;
    uint32_t idx = 0;
    for (const char* r : rows) {
        std::stringstream ss(r);
        std::string a, b, c;
        ss >> a >> b >> c;
        HostSearch s;
        auto val = [](char ch) -> uint32_t { return ch >= 'a' ? (uint32_t)(ch - 'a' + 10) : (uint32_t)(ch - '0'); }; // (a = 10 ...)
        for (char ch : a) s.pi.push_back(val(ch));
        for (char ch : b) s.L.push_back(val(ch));
        for (char ch : c) s.U.push_back(val(ch));
        s.sIdx = idx++;
        sch.searches.push_back(s);
    }
    sch.finalize();
    return sch;
}

// SearchScheme::mirrorPiStrings (search.h:488-493, :745-753): part i becomes part P - 1 - i in every search
inline HostScheme mirrored(const HostScheme& in) {
    HostScheme out;
    out.k = in.k;
    for (const HostSearch& h : in.searches) {
        HostSearch m = h;
        for (auto& p : m.pi) p = (uint32_t)h.pi.size() - 1 - p;
        out.searches.push_back(m);
    }
    out.finalize();
    return out;
}

// MinUSearchStrategy (searchstrategy.h:3284-3389), k = 1..7
inline HostScheme minU(uint32_t k) {
    switch (k) {
    case 1: return schemeFromRows(1, {"01 00 01", "10 00 01"});
    case 2: return schemeFromRows(2, {"012 011 022", "102 000 012", "210 002 012"});
    case 3: return schemeFromRows(3, {"0123 0000 0133", "1023 0111 0133", "2310 0002 0133", "3210 0113 0133"});
    case 4:
        return schemeFromRows(4, {"01234 00222 02244", "12034 00000 01244", "21034 01111 01244", "34210 00003 01444",
                                  "43210 01114 01444"});
    case 5:
        return schemeFromRows(5, {"012345 000222 013555", "102345 011333 013555", "231045 000000 013355",
                                  "321045 011111 013355", "453210 000004 013555", "543210 011115 013555"});
    case 6:
        return schemeFromRows(6, {"0123456 0022226 0226666", "1203456 0111115 0126666", "2103456 0000004 0126666",
                                  "3456210 0000000 0133666", "4356210 0111111 0133666", "5643210 0002222 0133666",
                                  "6543210 0113333 0133666"});
    case 7:
        return schemeFromRows(7, {"01234567 00000000 01337777", "10234567 01111111 01337777", "23104567 00022222 01337777",
                                  "32104567 01133333 01337777", "45673210 00000004 01337777", "54673210 01111115 01337777",
                                  "67543210 00022226 01337777", "76543210 01133337 01337777"});
    default: throw std::runtime_error("minU schemes exist for 1 to 7 errors");
    }
}

// ColumbaSearchStrategy's greedy schemes for 8 .. 13 errors (searchstrategy.h:3417-3658: ED8 .. ED13; k + 1 parts, k + 1
// searches; digits beyond 9 are written a, b, c, d).  The rows are the reference's data files search_schemes/pigeon_adapted/<k>/
// searches.txt (tools/gen_greedy_schemes.py), which hold the same tables as the class source.
inline HostScheme greedy(uint32_t k) {
    switch (k) {
    case 8:
        return schemeFromRows(8, {"012345678 012345678 023578888", "102345678 001234567 014578888",
                                  "210345678 000001234 012678888", "321045678 000112345 013378888",
                                  "432105678 000023456 013448888", "543210678 000000012 012555888",
                                  "654321078 000000123 012566688", "765432108 000000000 012377778",
                                  "876543210 000000001 012388888"});
    case 9:
        return schemeFromRows(9, {"0123456789 0000000000 0123499999", "1234567890 0000000012 0123888889",
                                  "2345678910 0000000001 0123777799", "3456789210 0000001234 0126666999",
                                  "4567893210 0000000123 0125559999", "5678943210 0000345678 0144499999",
                                  "6789543210 0000012345 0133799999", "7896543210 0023456789 0225799999",
                                  "8976543210 0001123456 0135799999", "9876543210 0112234567 0135799999"});
    case 10:
        return schemeFromRows(10, {"0123456789a 0123456789a 023579aaaaa", "1023456789a 00123456789 014579aaaaa",
                                  "2103456789a 00000123456 012679aaaaa", "3210456789a 00011234567 013379aaaaa",
                                  "4321056789a 00002345678 013449aaaaa", "5432106789a 00000001234 012555aaaaa",
                                  "6543210789a 00000012345 0125666aaaa", "7654321089a 00000000012 01237777aaa",
                                  "8765432109a 00000000123 012388888aa", "9876543210a 00000000000 0123499999a",
                                  "a9876543210 00000000001 01234aaaaaa"});
    case 11:
        return schemeFromRows(11, {"0123456789ab 000000000000 012345bbbbbb", "123456789ab0 000000000012 01234aaaaaab",
                                  "23456789ab10 000000000001 0123499999bb", "3456789ab210 000000001234 012388888bbb",
                                  "456789ab3210 000000000123 01237777bbbb", "56789ab43210 000000123456 0125666bbbbb",
                                  "6789ab543210 000000012345 012555bbbbbb", "789ab6543210 000023456789 013449bbbbbb",
                                  "89ab76543210 000112345678 013379bbbbbb", "9ab876543210 000001234567 012679bbbbbb",
                                  "ab9876543210 00123456789a 014579bbbbbb", "ba9876543210 0123456789ab 023579bbbbbb"});
    case 12:
        return schemeFromRows(12, {"0123456789abc 0123456789abc 023579bcccccc", "1023456789abc 00123456789ab 014579bcccccc",
                                  "2103456789abc 0000012345678 012679bcccccc", "3210456789abc 0001123456789 013379bcccccc",
                                  "4321056789abc 000023456789a 013449bcccccc", "5432106789abc 0000000123456 012555bcccccc",
                                  "6543210789abc 0000001234567 0125666cccccc", "7654321089abc 0000000001234 01237777ccccc",
                                  "8765432109abc 0000000012345 012388888cccc", "9876543210abc 0000000000012 0123499999ccc",
                                  "a9876543210bc 0000000000123 01234aaaaaacc", "ba9876543210c 0000000000000 012345bbbbbbc",
                                  "cba9876543210 0000000000001 012345ccccccc"});
    case 13:
        return schemeFromRows(13, {"0123456789abcd 00000000000000 0123456ddddddd", "123456789abcd0 00000000000012 012345cccccccd",
                                  "23456789abcd10 00000000000001 012345bbbbbbdd", "3456789abcd210 00000000001234 01234aaaaaaddd",
                                  "456789abcd3210 00000000000123 0123499999dddd", "56789abcd43210 00000000123456 012388888ddddd",
                                  "6789abcd543210 00000000012345 01237777dddddd", "789abcd6543210 00000012345678 0125666ddddddd",
                                  "89abcd76543210 00000001234567 012555bddddddd", "9abcd876543210 000023456789ab 013449bddddddd",
                                  "abcd9876543210 0001123456789a 013379bddddddd", "bcda9876543210 00000123456789 012679bddddddd",
                                  "cdba9876543210 00123456789abc 014579bddddddd", "dcba9876543210 0123456789abcd 023579bddddddd"});
    default: throw std::runtime_error("the greedy schemes exist for 8 to 13 errors");
    }
}

inline void fillNamed(cmb_strategy& st, const std::string& name) {
    if (name == "kuch1") { // KucherovKPlus1 (searchstrategy.h:2829-2913)
        st.kmerCutOff = 100;
        st.schemes[1] = {schemeFromRows(1, {"01 01 01", "10 00 01"})};
        st.schemes[2] = {schemeFromRows(2, {"012 000 022", "210 000 012", "102 001 012"})};
        st.schemes[3] = {schemeFromRows(3, {"0123 0000 0133", "1023 0011 0133", "2310 0000 0133", "3210 0011 0133"})};
        st.schemes[4] = {schemeFromRows(4, {"01234 00000 02244", "43210 00000 01344", "10234 00133 01334",
                                            "01234 00133 01334", "32410 00011 01244", "21034 00013 01244",
                                            "10234 00124 01244", "01234 00034 00444"})};
        st.params[1] = PartitionParams{{}, {0.5}, {1, 1}};
        st.params[2] = PartitionParams{{0.57}, {0.41, 0.7}, {39, 10, 40}};
        st.params[3] = PartitionParams{{0.38, 0.65}, {0.25, 0.50, 0.75}, {400, 4, 5, 400}};
        st.params[4] = PartitionParams{{0.38, 0.55, 0.73}, {0.27, 0.47, 0.62, 0.81}, {100, 5, 1, 6, 105}};
    } else if (name == "pigeon") { // PigeonHoleSearchStrategy (searchstrategy.h:3221-3274)
        st.kmerCutOff = 20;
        st.schemes[1] = {schemeFromRows(1, {"01 00 01", "10 00 01"})};
        st.schemes[2] = {schemeFromRows(2, {"012 000 022", "120 000 022", "210 000 022"})};
        st.schemes[3] = {schemeFromRows(3, {"0123 0000 0333", "1023 0000 0333", "2310 0000 0333", "3210 0000 0333"})};
        st.schemes[4] = {schemeFromRows(4, {"01234 00000 04444", "12340 00000 04444", "23410 00000 04444",
                                            "34210 00000 04444", "43210 00000 04444"})};
    } else if (name == "multiple_opt") { // search_schemes/multiple_opt/{2,4,6}/scheme<i>.txt
        st.kmerCutOff = 20;
        st.schemes[2] = {schemeFromRows(2, {"012 011 022", "102 000 012", "210 002 012"}),
                         schemeFromRows(2, {"210 011 022", "120 000 012", "012 002 012"})};
        st.schemes[4] = {
            schemeFromRows(4, {"01234 00222 02244", "12034 00000 01244", "21034 01111 01244", "34210 00003 01444",
                               "43210 01114 01444"}),
            schemeFromRows(4, {"01234 01114 01444", "10234 00003 01444", "23410 01111 02244", "32410 00000 01244",
                               "43210 00222 01244"}),
            schemeFromRows(4, {"43210 00222 02244", "32410 00000 01244", "23410 01111 01244", "10234 00003 01444",
                               "01234 01114 01444"})};
        st.schemes[6] = {
            schemeFromRows(6, {"0123456 0022226 0226666", "1203456 0111115 0126666", "2103456 0000004 0126666",
                               "3456210 0000000 0133666", "4356210 0111111 0133666", "5643210 0002222 0133666",
                               "6543210 0113333 0133666"}),
            schemeFromRows(6, {"0123456 0111115 0126666", "1023456 0000004 0126666", "2103456 0022226 0226666",
                               "3456210 0002222 0133666", "4356210 0113333 0133666", "5643210 0000000 0133666",
                               "6543210 0111111 0133666"}),
            schemeFromRows(6, {"6543210 0111115 0126666", "5643210 0000004 0126666", "4563210 0022226 0226666",
                               "3210456 0002222 0133666", "2310456 0113333 0133666", "1023456 0000000 0133666",
                               "0123456 0111111 0133666"}),
            schemeFromRows(6, {"6543210 0022226 0226666", "5463210 0111115 0126666", "4563210 0000004 0126666",
                               "3210456 0000000 0133666", "2310456 0111111 0133666", "1023456 0002222 0133666",
                               "0123456 0113333 0133666"})};
    } else if (name == "kuch2") { // KucherovKPlus2 (searchstrategy.h:2918-3021)
        st.kmerCutOff = 100;
        st.schemes[1] = {schemeFromRows(1, {"012 000 011", "120 000 001"})};
        st.schemes[2] = {schemeFromRows(2, {"0123 0000 0112", "3210 0000 0122", "1230 0001 0012", "0123 0002 0022"})};
        st.schemes[3] = {schemeFromRows(3, {"01234 00000 01233", "12340 00000 01223", "23410 00001 01133", "34210 00012 00333"})};
        st.schemes[4] = {schemeFromRows(4, {"012345 000000 012344", "123450 000000 012344", "543210 000001 012244",
                                            "345210 000012 011344", "234510 000023 011244", "453210 000133 003344",
                                            "012345 000333 003344", "012345 000044 002444", "231045 000124 002244",
                                            "453210 000044 001444"})};
        st.params[1] = PartitionParams{{0.94}, {0.47, 0.94}, {11, 10, 1}};
        st.params[2] = PartitionParams{{0.48, 0.55}, {0.35, 0.50, 0.65}, {400, 4, 1, 800}};
        st.params[3] = PartitionParams{{0.4, 0.63, 0.9}, {0.22, 0.44, 0.66, 0.88}, {6, 3, 2, 1, 1}};
        st.params[4] = PartitionParams{{0.34, 0.5, 0.65, 0.7}, {0.18, 0.37, 0.53, 0.69, 0.83}, {52, 42, 16, 14, 1, 800}};
    } else if (name == "kianfar") { // OptimalKianfar (searchstrategy.h:3026-3103)
        st.kmerCutOff = 100;
        st.schemes[1] = {schemeFromRows(1, {"01 00 01", "10 01 01"})};
        st.schemes[2] = {schemeFromRows(2, {"012 002 012", "210 000 022", "120 011 012"})};
        st.schemes[3] = {schemeFromRows(3, {"0123 0003 0233", "1230 0000 1233", "2310 0022 0033"})};
        st.schemes[4] = {schemeFromRows(4, {"01234 00004 03344", "12340 00000 22334", "43210 00033 00444"})};
        st.params[1] = PartitionParams{{}, {0.5}, {1, 1}};
        st.params[2] = PartitionParams{{0.50}, {0.30, 0.60}, {10, 1, 5}};
        st.params[3] = PartitionParams{{0.34, 0.66}, {0.17, 0.69, 0.96}, {1, 1, 1, 1}};
        st.params[4] = PartitionParams{{0.42, 0.56, 0.67}, {0.2, 0.5, 0.6, 0.8}, {7, 2, 1, 3, 5}};
    } else if (name == "01*0") { // O1StarSearchStrategy (searchstrategy.h:3115-3206)
        st.kmerCutOff = 100;
        st.schemes[1] = {schemeFromRows(1, {"012 000 011", "120 000 001"})};
        st.schemes[2] = {schemeFromRows(2, {"0123 0000 0122", "1230 0000 0122", "2310 0000 0022"})};
        st.schemes[3] = {schemeFromRows(3, {"01234 00000 01333", "12340 00000 01333", "23410 00000 01333", "34210 00000 00333"})};
        st.schemes[4] = {schemeFromRows(4, {"012345 000000 014444", "123450 000000 014444", "234510 000000 014444",
                                            "345210 000000 014444", "453210 000000 004444"})};
        st.params[1] = PartitionParams{{0.94}, {0.50, 0.96}, {11, 10, 1}};
        st.params[2] = PartitionParams{{0.51, 0.93}, {0.26, 0.64, 0.83}, {20, 11, 11, 10}};
        st.params[3] = PartitionParams{{0.34, 0.64, 0.88}, {0.22, 0.46, 0.67, 0.95}, {3, 2, 2, 1, 1}};
        st.params[4] = PartitionParams{{0.28, 0.48, 0.63, 0.94}, {0.19, 0.37, 0.57, 0.74, 0.96}, {1, 2, 2, 1, 2, 1}};
    } else if (name == "minU") { // MinUSearchStrategy (searchstrategy.h:3284-3389): base-class partition defaults
        st.kmerCutOff = 20;
        for (uint32_t k = 1; k <= 7; k++) st.schemes[k] = {minU(k)};
    } else if (name == "columba") {
        // `-S columba`, the CLI default: DynamicColumbaStrategy (searchstrategy.h:3666-3736) = for every k the scheme of
        // ColumbaSearchStrategy (minU up to 7 errors, the greedy schemes for 8 .. 13 errors) and its mirror image
        // (MultipleSchemes ctor, :2468-2477), then the "middle" schemes for k = 2, 4, 6 (+ the mirror of the k = 6 one);
        // base-class partition defaults.  (8 .. 13 errors: Hamming-distance batches run on the wide device tables, MAXP_WIDE parts;
        // edit-distance batches likewise: wide record geometry of the frontier, from 11 errors on the in-index matrix with 16-row blocks (GeoX).)
        st.kmerCutOff = 20;
        for (uint32_t k = 1; k <= 13; k++) {
            const HostScheme m = k <= 7 ? minU(k) : greedy(k);
            st.schemes[k] = {m, mirrored(m)};
        }
        st.schemes[2].push_back(schemeFromRows(2, {"210 011 022", "120 000 012", "012 002 012"}));
        st.schemes[4].push_back(schemeFromRows(4, {"01234 01114 01444", "10234 00003 01444", "23410 01111 02244",
                                                   "32410 00000 01244", "43210 00222 01244"}));
        const HostScheme mid6 = schemeFromRows(6, {"0123456 0111115 0126666", "1023456 0000004 0126666", "2103456 0022226 0226666",
                                                   "3456210 0002222 0133666", "4356210 0113333 0133666", "5643210 0000000 0133666",
                                                   "6543210 0111111 0133666"});
        st.schemes[6].push_back(mid6);
        st.schemes[6].push_back(mirrored(mid6));
    } else if (name == "naive") {
        // `-S naive`: NaiveBackTrackingStrategy (searchstrategy.h:2785-2820) — ONE part for every k, so partition() never splits
        // (searchstrategy.cpp:148-152) and every read is matched by IndexInterface::approxMatchesNaive[Hamming] (:442-459).  The
        // single search (0)(0)(k) the class lists is never run.
        st.kmerCutOff = 20;
        for (uint32_t k = 1; k <= 13; k++) {
            const std::string row = std::string("0 0 ") + (char)(k < 10 ? '0' + k : 'a' + (k - 10));
            st.schemes[k] = {schemeFromRows(k, {row.c_str()})};
        }
    } else {
        throw std::runtime_error(name + " is not an option as search scheme");
    }
}

inline bool fileExists(const std::string& p) {
    std::ifstream f(p);
    return f.good();
}

// `-d <dir>`: <dir>/<k>/scheme<i>.txt (searchstrategy.h:2378-2410, :2624-2660); base-class
// partition defaults, k-mer cut-off 20.
inline void fillFromMultipleDir(cmb_strategy& st, std::string dir) {
    if (!dir.empty() && dir.back() != '/') dir += '/';
    if (!fileExists(dir + "name.txt"))
        throw std::runtime_error("Problem reading: " + dir +
                                 "name.txt\nDid you provide a directory to a search scheme without a name file?");
    st.kmerCutOff = 20;
    for (uint32_t k = 1; k <= 13; k++) { // MAX_K (definitions.h:50)
        std::vector<HostScheme> alts;
        for (int x = 1;; x++) {
            const std::string p = dir + std::to_string(k) + "/scheme" + std::to_string(x) + ".txt";
            if (!fileExists(p)) break;
            alts.push_back(readSchemeFile(p, k));
        }
        if (!alts.empty()) st.schemes[k] = alts;
    }
}

// `-c <dir>`: <dir>/<k>/searches.txt (+ optional static_partitioning.txt, dynamic_partitioning.txt); k-mer cut-off
// 50 (searchstrategy.h:2308).  dynamicSelection = false is `-c <dir> -nD` (CustomSearchStrategy); true is the CLI's
// default for `-c` (DynamicCustomStrategy, searchstrategy.h:3744-3776): every scheme plus its mirror image with
// dynamic selection — that strategy object is a COPY of the base class (MultipleSchemesStrategy(SearchStrategy*),
// :2689), so the partitioning files of the directory no longer apply (base-class defaults), the cut-off stays 50.
inline void fillFromCustomDir(cmb_strategy& st, std::string dir, bool dynamicSelection = false) {
    if (!dir.empty() && dir.back() != '/') dir += '/';
    if (!fileExists(dir + "name.txt"))
        throw std::runtime_error("Problem reading: " + dir +
                                 "name.txt\nDid you provide a directory to a search scheme without a name file?");
    st.kmerCutOff = 50;
    for (uint32_t k = 1; k <= 13; k++) {
        const std::string base = dir + std::to_string(k) + "/";
        if (!fileExists(base + "searches.txt")) continue;
        st.schemes[k] = {readSchemeFile(base + "searches.txt", k)};
        if (dynamicSelection) {
            st.schemes[k].push_back(mirrored(st.schemes[k].front()));
            continue;
        }
        PartitionParams pp;
        const uint32_t P = st.schemes[k].front().numParts();
        auto tokensOfLine = [](std::ifstream& f) {
            std::string line, t;
            std::getline(f, line);
            std::stringstream ss(line);
            std::vector<std::string> v;
            while (ss >> t) v.push_back(t);
            return v;
        };
        // first line of static_partitioning.txt: P - 1 begin positions (searchstrategy.cpp:2033-2063, :2166-2183)
        {
            std::ifstream f(base + "static_partitioning.txt");
            if (f) {
                const auto toks = tokensOfLine(f);
                if (toks.size() != P - 1)
                    throw std::runtime_error("Not enough static positions provided in " + base +
                                             "static_partitioning.txt\nExpected: " + std::to_string(P - 1) +
                                             " parts\nProvided: " + std::to_string(toks.size()) + " parts");
                for (const auto& t : toks) pp.begins.push_back(std::stod(t));
                for (size_t i = 0; i < pp.begins.size(); i++) {
                    if (pp.begins[i] <= 0 || pp.begins[i] >= 1)
                        throw std::runtime_error("One of the provided static positions for " + std::to_string(k) +
                                                 " is not between 0 and 1 (exclusive)");
                    if (i + 1 < pp.begins.size() && pp.begins[i] - pp.begins[i + 1] >= 0)
                        throw std::runtime_error("Provided static positions for " + std::to_string(k) +
                                                 " are not strictly increasing");
                }
            }
        }
        // dynamic_partitioning.txt: P - 2 seeding positions, then P weights (searchstrategy.cpp:2066-2117, :2184-2203)
        {
            std::ifstream f(base + "dynamic_partitioning.txt");
            if (f) {
                const auto toks = tokensOfLine(f);
                if (toks.size() != P - 2)
                    throw std::runtime_error("Not enough seeding positions provided in " + base +
                                             "dynamic_partitioning.txt\nExpected: " + std::to_string(P - 1) +
                                             " seeds\nProvided: " + std::to_string(toks.size()) + " seeds");
                for (const auto& t : toks) pp.seeding.push_back(std::stod(t));
                for (size_t i = 0; i < pp.seeding.size(); i++) {
                    if (pp.seeding[i] <= 0 || pp.seeding[i] >= 1)
                        throw std::runtime_error("One of the provided static positions for " + std::to_string(k) +
                                                 " is not between 0 and 1 (exclusive)!");
                    if (i + 1 < pp.seeding.size() && pp.seeding[i] - pp.seeding[i + 1] >= 0)
                        throw std::runtime_error("Provided seeding positions for " + std::to_string(k) +
                                                 " are not strictly increasing");
                }
                for (const auto& t : tokensOfLine(f)) pp.weights.push_back((uint64_t)std::stoi(t));
                if (pp.weights.size() != P)
                    throw std::runtime_error("Not enough weights provided for max score " + std::to_string(k));
            }
        }
        st.params[k] = pp;
    }
}

} // namespace cmb

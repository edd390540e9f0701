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

inline DevSearch toDevSearch(const HostSearch& h) {
    const size_t n = h.pi.size();
    if (n != h.L.size() || n != h.U.size())
        throw std::runtime_error("Could not create search, the sizes of all vectors are not equal");
    if (n < 2 || n > (size_t)MAXP) throw std::runtime_error("unsupported number of parts in search");
    DevSearch d{};
    d.n = (uint8_t)n;
    for (size_t i = 0; i < n; i++) {
        d.order[i] = (uint8_t)h.pi[i];
        d.L[i] = (uint8_t)h.L[i];
        d.U[i] = (uint8_t)h.U[i];
    }
    // directions: phase 0 copies phase 1 (search.h:131)
    d.dir[0] = h.pi[1] > h.pi[0] ? 0 : 1;
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
        if (toks.empty()) continue;
        if (toks.size() != 3)
            throw std::runtime_error("Something went wrong with processing line: " + line + "\nin file: " + path +
                                     "\nA search should have 3 vectors: order, lower bound and upper bound!");
        HostSearch s;
        s.pi = parseBraces(toks[0]);
        s.L = parseBraces(toks[1]);
        s.U = parseBraces(toks[2]);
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

    // flatten everything matchWithSearches needs for distance k
    cmb::DevStrategyK flatten(uint32_t k) const {
        using namespace cmb;
        auto it = schemes.find(k);
        if (it == schemes.end() || it->second.empty())
            throw std::runtime_error("the search strategy does not support distance " + std::to_string(k));
        const auto& alts = it->second;
        if (alts.size() > (size_t)MAXSCH) throw std::runtime_error("too many alternative schemes");
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
            for (size_t i = 0; i < h.searches.size(); i++) d.sch[a].s[i] = toDevSearch(h.searches[i]);
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
        for (char ch : a) s.pi.push_back(ch - '0');
        for (char ch : b) s.L.push_back(ch - '0');
        for (char ch : c) s.U.push_back(ch - '0');
        s.sIdx = idx++;
        sch.searches.push_back(s);
    }
    sch.finalize();
    return sch;
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

// `-c <dir>` without dynamic selection: <dir>/<k>/searches.txt (+ optional static_partitioning.txt,
// dynamic_partitioning.txt); k-mer cut-off 50 (searchstrategy.h:2308)
inline void fillFromCustomDir(cmb_strategy& st, std::string dir) {
    if (!dir.empty() && dir.back() != '/') dir += '/';
    if (!fileExists(dir + "name.txt"))
        throw std::runtime_error("Problem reading: " + dir +
                                 "name.txt\nDid you provide a directory to a search scheme without a name file?");
    st.kmerCutOff = 50;
    for (uint32_t k = 1; k <= 13; k++) {
        const std::string base = dir + std::to_string(k) + "/";
        if (!fileExists(base + "searches.txt")) continue;
        st.schemes[k] = {readSchemeFile(base + "searches.txt", k)};
        PartitionParams pp;
        {
            std::ifstream f(base + "static_partitioning.txt");
            double v;
            while (f >> v) pp.begins.push_back(v);
        }
        {
            std::ifstream f(base + "dynamic_partitioning.txt");
            std::string l1, l2;
            if (f) {
                std::getline(f, l1);
                std::getline(f, l2);
                std::stringstream a(l1), b(l2);
                double v;
                while (a >> v) pp.seeding.push_back(v);
                uint64_t w;
                while (b >> w) pp.weights.push_back(w);
            }
        }
        st.params[k] = pp;
    }
}

} // namespace cmb

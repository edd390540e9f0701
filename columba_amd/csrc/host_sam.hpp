// Host side of the output records: read clean-up, CIGAR strings and SAM lines of single-end
// reads, formatted exactly as the reference formats them.
//
// Mirrors (reference, src/):
//   Read::cleanUpRecord / ReadBundle            reads.h:43-58, :97-160   (sequence id, upper case, non-ACGT -> N,
//                                                                         reverse complement, reversed quality)
//   TextOcc::getFlagsSE / getMapQ / asXA         indexhelpers.h:321-331, :378-388, :416-421
//   TextOcc::generateSAMSingleEnd / ...XA        indexhelpers.cpp:56-120
//   TextOcc::createUnmappedSAMOccurrenceSE       indexhelpers.cpp:177-200
//   SearchStrategy::generateOutputSingleEnd      searchstrategy.cpp:1824-1902 (primary = first occurrence of minimal
//                                                                              distance, nHits = how many share it)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace cmb {

constexpr int MAX_MAPQ = 60; // definitions.h:49

inline std::string cleanSeqID(std::string id) { // reads.h:43-52
    const size_t sp = id.find(' ');
    if (sp != std::string::npos) id.erase(sp);
    return id.empty() ? id : id.substr(1);
}
inline std::string cleanReadSeq(std::string s) { // reads.h:54-58, :97-101
    for (auto& c : s) {
        c = (char)std::toupper((unsigned char)c);
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') c = 'N';
    }
    return s;
}
inline std::string revComplWithN(const std::string& s) { // nucleotide.h:250
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); i++) {
        const char c = s[s.size() - 1 - i];
        r[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
    }
    return r;
}

inline std::string cigarString(const uint16_t* ops, uint32_t n) { // "57M1I92M" from (length << 2 | op) runs
    std::string s;
    for (uint32_t i = 0; i < n; i++) {
        s += std::to_string((unsigned)(ops[i] >> 2));
        s += "MID?"[ops[i] & 3u];
    }
    return s;
}

inline int mapQ(uint32_t distance, uint32_t nHits, uint32_t minScore) { // indexhelpers.h:378-388
    if (distance != minScore) return 0;
    if (nHits == 1) return MAX_MAPQ;
    return (int)std::round(-10.0 * std::log10(1 - 1.0 / nHits));
}

struct SamHit {
    std::string seqName, cigar;
    uint32_t pos0 = 0, distance = 0; // 0-based begin inside the sequence
    bool revCompl = false;
};

// indexhelpers.cpp:56-91
inline std::string samLineSE(const std::string& seqID, const SamHit& h, bool primary, uint32_t nHits, uint32_t minScore,
                             const std::string& printSeq, const std::string& printQual) {
    const unsigned flags = (h.revCompl ? 16u : 0u) | (primary ? 0u : 256u);
    std::string o;
    o.reserve(seqID.size() + printSeq.size() + printQual.size() + 100);
    o += seqID;
    o += '\t';
    o += std::to_string(flags);
    o += '\t';
    o += h.seqName;
    o += '\t';
    o += std::to_string(h.pos0 + 1); // SAM is 1-based
    o += '\t';
    o += std::to_string(mapQ(h.distance, nHits, minScore));
    o += '\t';
    o += h.cigar;
    o += "\t*\t0\t0\t";
    o += printSeq;
    o += '\t';
    o += printQual;
    o += "\tAS:i:";
    o += std::to_string(h.distance);
    o += "\tNM:i:";
    o += std::to_string(h.distance);
    o += "\tPG:Z:Columba\n";
    return o;
}
// indexhelpers.cpp:93-120 + indexhelpers.h:416-421, :647-663: the first hit's line with the others in the XA tag
inline std::string samLineSEWithXA(const std::string& seqID, const std::vector<SamHit>& hits, uint32_t nHits,
                                   const std::string& printSeq, std::string printQual) {
    if (printQual.empty()) printQual = "*";
    std::string o = samLineSE(seqID, hits.front(), true, nHits, hits.front().distance, printSeq, printQual);
    o.pop_back();
    const uint32_t x0 = nHits - 1;
    const uint32_t x1 = (uint32_t)(hits.size() - 1) - x0;
    o += "\tX0:i:" + std::to_string(x0) + "\tX1:i:" + std::to_string(x1) + "\tXA:Z:";
    for (size_t i = 1; i < hits.size(); i++) {
        const SamHit& h = hits[i];
        o += h.seqName + "," + (h.revCompl ? "-" : "+") + std::to_string(h.pos0 + 1) + "," + h.cigar + "," +
             std::to_string(h.distance) + ";";
    }
    o += "\n";
    return o;
}
// indexhelpers.cpp:177-200
inline std::string samLineUnmappedSE(const std::string& seqID, const std::string& read, const std::string& qual) {
    return seqID + "\t4\t*\t0\t0\t*\t*\t0\t0\t" + read + "\t" + qual + "\tPG:Z:Columba\n";
}

} // namespace cmb

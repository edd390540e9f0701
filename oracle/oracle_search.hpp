// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_core.hpp header).
//
// Restatement of the search layer: IndexInterface DFS (indexinterface.cpp),
// MatrixMetaInfo cluster (indexhelpers.{h,cpp}), in-text verification
// (fmindex.cpp, indexhelpers.cpp:518), SearchStrategy partitioning and drivers
// (searchstrategy.{h,cpp}).  PARITY UNPINNED against a reference binary for
// this layer (parallel_hashmap is absent from the image, so the reference
// translation units cannot be built without a stand-in); every function cites
// the lines it follows.
// ============================================================================
#pragma once
#include <atomic>
#include "oracle_core.hpp"

namespace orc {
// the 32-bit narrow-block experiment (ORC_NARROW_BLOCKS=32): phases run on the 32-bit matrix / phases that fell back to the reference's
inline std::atomic<uint64_t> g_narrow32Stats[2];

// ----------------------------------------------------------------------------
// Occurrence value types (indexhelpers.h:289, :1283, :1353, :1544)
// ----------------------------------------------------------------------------
struct TextOcc {
    Range range;
    len_t distance = 0;
    Strand strand = FORWARD_STRAND;
    std::vector<std::pair<char, uint32_t>> cigar; // empty == no CIGAR
    TextOcc() {}
    TextOcc(Range r, len_t d, Strand s) : range(r), distance(d), strand(s) {}
    bool hasCigar() const { return !cigar.empty(); }
    // indexhelpers.h:779-795
    bool operator<(const TextOcc& r) const {
        if (range.b != r.range.b) return range.b < r.range.b;
        if (distance != r.distance) return distance < r.distance;
        if (range.width() != r.range.width()) return range.width() < r.range.width();
        return hasCigar() && !r.hasCigar();
    }
    // indexhelpers.h:811
    bool operator==(const TextOcc& r) const {
        return r.range == range && r.distance == distance;
    }
};

// The types below are templates on the range-pair type RP: orc::RangePair for the FM-index (Vanilla flavour), the move
// ranges with toehold of the run-length compressed flavour (indexhelpers.h:1117-1260 under RUN_LENGTH_COMPRESSION,
// oracle_move.hpp: MovePairT) — the search layer is the same code in the reference, compiled twice.
inline len_t saBeginOf(const RangePair& r) { return r.sa.b; }

template <class RP> struct FMPosT {
    RP ranges;
    len_t depth = 0;
    FMPosT() {}
    FMPosT(const RP& r, len_t d) : ranges(r), depth(d) {}
    bool isValid() const { return !ranges.empty(); }
};

template <class RP> struct FMOccT {
    FMPosT<RP> pos;
    len_t distance = 0;
    len_t shift = 0;
    Strand strand = FORWARD_STRAND;
    FMOccT() {}
    FMOccT(const RP& r, len_t dist, len_t depth, Strand s = FORWARD_STRAND,
          len_t sh = 0)
        : pos(r, depth), distance(dist), shift(sh), strand(s) {}
    const RP& getRanges() const { return pos.ranges; }
    len_t getDepth() const { return pos.depth; }
    bool isValid() const { return pos.isValid(); }
    // indexhelpers.h:1506-1524
    bool operator<(const FMOccT& rhs) const {
        if (saBeginOf(pos.ranges) != saBeginOf(rhs.pos.ranges)) return saBeginOf(pos.ranges) < saBeginOf(rhs.pos.ranges);
        if (distance != rhs.distance) return distance < rhs.distance;
        if (pos.ranges.width() != rhs.pos.ranges.width())
            return pos.ranges.width() < rhs.pos.ranges.width();
        return shift < rhs.shift;
    }
    // indexhelpers.h:1529
    bool operator==(const FMOccT& rhs) const {
        return getRanges() == rhs.getRanges() && distance == rhs.distance &&
               getDepth() == rhs.getDepth() && shift == rhs.shift && strand == rhs.strand;
    }
};

template <class RP> struct FMPosExtT : FMPosT<RP> {
    char c = 0;
    bool reported = false;
    FMPosExtT() {}
    FMPosExtT(char ch, const RP& r, len_t row) : FMPosT<RP>(r, row), c(ch) {}
    len_t getRow() const { return this->depth; }
    // indexhelpers.h:1586-1601
    void report(FMOccT<RP>& occ, len_t startDepth, len_t EDFound, bool noDoubleReports,
                len_t shift) {
        if (!reported) {
            occ = FMOccT<RP>(this->ranges, EDFound, this->depth + startDepth, FORWARD_STRAND, shift);
            if (noDoubleReports) reported = true;
        }
    }
};

template <class RP> struct OccurrencesT { // indexhelpers.h:1957
    typedef FMOccT<RP> FMOcc;
    std::vector<TextOcc> inTextOcc;
    std::vector<FMOcc> inFMOcc;
    void eraseDoublesFM() { // :2135 (DEVELOPER_MODE: stable)
        std::stable_sort(inFMOcc.begin(), inFMOcc.end());
        inFMOcc.erase(std::unique(inFMOcc.begin(), inFMOcc.end()), inFMOcc.end());
    }
    void eraseDoublesAndSortText() { // :2148
        std::stable_sort(inTextOcc.begin(), inTextOcc.end());
        inTextOcc.erase(std::unique(inTextOcc.begin(), inTextOcc.end()), inTextOcc.end());
    }
};

// ----------------------------------------------------------------------------
// MatrixMetaInfo — the final-column cluster (indexhelpers.h:1677-1838,
// indexhelpers.cpp:276-382)
// ----------------------------------------------------------------------------
template <class RP> struct ClusterT {
    typedef FMOccT<RP> FMOcc;
    typedef FMPosExtT<RP> FMPosExt;
    std::vector<uint16_t> eds;
    std::vector<FMPosExt> nodes;
    len_t lastCell;
    uint16_t maxED;
    len_t startDepth, shift;
    ClusterT(len_t size, len_t maxED_, len_t startDepth_, len_t shift_)
        : eds(size, (uint16_t)(maxED_ + 1)), nodes(size), lastCell((len_t)-1),
          maxED((uint16_t)maxED_), startDepth(startDepth_), shift(shift_) {}
    void setValue(len_t idx, const FMPosExt& node, len_t ed) {
        eds[idx] = (uint16_t)ed;
        nodes[idx] = node;
        lastCell = idx;
    }
    len_t size() const { return (len_t)eds.size(); }
    // indexhelpers.h:1743-1761
    std::vector<FMOcc> reportCentersAtEnd() {
        std::vector<FMOcc> centers;
        for (len_t i = 0; i <= lastCell && lastCell != (len_t)-1; i++) {
            if (eds[i] <= maxED && (i == 0 || eds[i] <= eds[i - 1]) &&
                (i == lastCell || eds[i] <= eds[i + 1])) {
                FMOcc m;
                nodes[i].report(m, startDepth, eds[i], true, shift);
                centers.emplace_back(m);
            }
        }
        return centers;
    }
    // indexhelpers.h:1770-1798
    FMOcc reportDeepestMinimum(Direction dir) {
        uint16_t minED = maxED + 1;
        len_t highestBestIdx = (len_t)-1, deepestBestIdx = (len_t)-1;
        for (len_t i = 0; i <= lastCell && lastCell != (len_t)-1; i++) {
            if (eds[i] < minED) {
                minED = eds[i];
                highestBestIdx = i;
                deepestBestIdx = i;
            }
            if (eds[i] == minED) deepestBestIdx = i;
        }
        FMOcc m;
        if (minED <= maxED) {
            nodes[deepestBestIdx].report(
                m, startDepth - (deepestBestIdx - highestBestIdx), minED, true,
                ((dir == BACKWARD) ? (deepestBestIdx - highestBestIdx) : 0) + shift);
        }
        return m;
    }
    // indexhelpers.cpp:276-382
    FMOcc getClusterCentra(uint16_t lowerBound, std::vector<FMPosExt>& desc,
                           std::vector<uint16_t>& initEds) {
        FMOcc m;
        for (len_t i = 0; i <= lastCell && lastCell != (len_t)-1; i++) {
            if (eds[i] > maxED || eds[i] < lowerBound) continue;
            bool betterThanParent = (i == 0) || eds[i] <= eds[i - 1];
            bool betterThanChild = (i == lastCell) || eds[i] <= eds[i + 1];
            if (betterThanParent && betterThanChild) {
                nodes[i].report(m, startDepth, eds[i], false, shift);
                initEds.emplace_back(eds[i]);
                for (len_t j = i + 1; j <= lastCell; j++) {
                    desc.emplace_back(nodes[j]);
                    initEds.emplace_back(eds[j]);
                }
                for (len_t k = 1; k < initEds.size(); k++) {
                    if (initEds[k] < lowerBound && initEds[k] <= initEds[k - 1] &&
                        (k == initEds.size() - 1 || initEds[k] <= initEds[k + 1])) {
                        len_t highestPoint = 0;
                        len_t lowestPoint = (len_t)initEds.size() - 1;
                        for (len_t l = k; l-- > 0;) {
                            if (initEds[l] != initEds[l + 1] + 1) {
                                highestPoint = l + 1;
                                break;
                            }
                        }
                        for (len_t l = k + 1; l < initEds.size(); l++) {
                            if (initEds[l] != initEds[l - 1] + 1) {
                                lowestPoint = l - 1;
                                break;
                            }
                        }
                        if (highestPoint != 0 && lowestPoint != initEds.size() - 1) {
                            len_t lC = lowestPoint, hC = highestPoint;
                            bool highest = true;
                            while (lC > hC) {
                                if (highest) {
                                    initEds[hC] = (uint16_t)std::min<int>(maxED + 1, initEds[hC - 1] + 1);
                                    hC++;
                                } else {
                                    initEds[lC] = (uint16_t)std::min<int>(maxED + 1, initEds[lC + 1] + 1);
                                    lC--;
                                }
                                highest = !highest;
                            }
                            if (lC == hC) {
                                initEds[lC] = (uint16_t)std::min<int>(initEds[lC + 1] + 1,
                                                                      initEds[lC - 1] + 1);
                            }
                        } else if (highestPoint == 0 && lowestPoint != initEds.size() - 1) {
                            for (len_t l = lowestPoint; l-- > 0;)
                                initEds[l] = initEds[l + 1] + 1;
                        } else if (highestPoint != 0 && lowestPoint == initEds.size() - 1) {
                            for (len_t l = highestPoint; l < initEds.size(); l++)
                                initEds[l] = initEds[l - 1] + 1;
                        }
                    }
                }
                break;
            }
        }
        return m;
    }
};

typedef FMPosT<RangePair> FMPos;
typedef FMOccT<RangePair> FMOcc;
typedef FMPosExtT<RangePair> FMPosExt;
typedef OccurrencesT<RangePair> Occurrences;
typedef ClusterT<RangePair> Cluster;

// ----------------------------------------------------------------------------
// Strategy description (searchstrategy.h: SearchStrategy + derived classes)
// ----------------------------------------------------------------------------
struct Strategy {
    DistanceMetric metric = EDIT;
    PartitionStrategy partition = DYNAMIC;
    len_t useKmerCutOff = 20; // searchstrategy.h:222 (100 for kuch1 :2907)
    // schemes[k] = list of alternative schemes for distance k (dynamic
    // selection, searchstrategy.h:2505-2537); one entry == fixed scheme
    std::vector<std::vector<SearchScheme>> schemesPerK; // index k (0 unused)
    // optional per-k overrides (empty -> base-class defaults)
    std::vector<std::vector<double>> seedingPositions; // :1825 / :2890
    std::vector<std::vector<uint64_t>> weights;        // :283 / :2885
    std::vector<std::vector<double>> begins;           // :245 / :2880

    len_t calculateNumParts(len_t k) const { return schemesPerK[k].front().getNumParts(); }
    bool supports(len_t k) const {
        return k < schemesPerK.size() && !schemesPerK[k].empty();
    }
    std::vector<double> getSeedingPositions(int numParts, int k) const {
        if ((size_t)k < seedingPositions.size() && !seedingPositions[k].empty()) return seedingPositions[k];
        double u = 1.0 / (numParts - 1);
        std::vector<double> s;
        for (int i = 1; i < numParts - 1; i++) s.push_back(i * u);
        return s;
    }
    std::vector<uint64_t> getWeights(int numParts, int k) const {
        if ((size_t)k < weights.size() && !weights[k].empty()) return weights[k];
        std::vector<uint64_t> w(numParts, 1);
        w.front() = 2;
        w.back() = 2;
        return w;
    }
    std::vector<double> getBegins(int numParts, int k) const {
        if ((size_t)k < begins.size() && !begins[k].empty()) return begins[k];
        std::vector<double> b;
        double u = 1.0 / numParts;
        for (int i = 1; i < numParts; i++) b.push_back(i * u);
        return b;
    }
    // MultipleSchemes::createSearches searchstrategy.h:2505-2537
    template <class RP> const std::vector<Search>& createSearches(len_t k, const std::vector<RP>& ranges) const {
        const auto& schemes = schemesPerK[k];
        if (schemes.size() == 1) return schemes[0].searches;
        unsigned numParts = schemes.front().getNumParts();
        unsigned int total = 0;
        for (const auto& r : ranges) total += r.width();
        if (total <= numParts) return schemes[0].searches;
        int minIndex = 0;
        unsigned int minValue = ranges[schemes[0].criticalPartIndex].width();
        for (unsigned i = 1; i < schemes.size(); ++i) {
            unsigned cp = schemes[i].criticalPartIndex;
            if (ranges[cp].width() < minValue) {
                minValue = ranges[cp].width();
                minIndex = (int)i;
            }
        }
        return schemes[minIndex].searches;
    }
};

// ----------------------------------------------------------------------------
// Matcher: per-thread state of IndexInterface + SearchStrategy
// ----------------------------------------------------------------------------
template <class IX> class MatcherT {
  public:
    typedef typename IX::RangePair RangePair; // (shadows orc::RangePair inside the matcher)
    typedef typename IX::SARange SARange;
    typedef FMPosT<RangePair> FMPos;
    typedef FMOccT<RangePair> FMOcc;
    typedef FMPosExtT<RangePair> FMPosExt;
    typedef OccurrencesT<RangePair> Occurrences;
    typedef ClusterT<RangePair> Cluster;
    static constexpr bool RLC = IX::RLC; // the reference's RUN_LENGTH_COMPRESSION branches
    const IX& index;
    const Strategy& strat;
    bool noCIGAR = true;
    Counters counters;

    MatcherT(const IX& idx, const Strategy& st) : index(idx), strat(st) {}

    // === SearchStrategy::matchApproxAllMap (searchstrategy.cpp:495-535) ===
    std::vector<TextOcc> matchApproxAll(const std::string& read, len_t maxED) {
        std::string rc = revCompl(read);
        std::vector<TextOcc> result;
        if (maxED == 0) {
            strand = FORWARD_STRAND;
            exactMatchesOutput(read, result);
            strand = REVERSE_C_STRAND;
            exactMatchesOutput(rc, result);
            return result;
        }
        Occurrences occ;
        fullReadMatrices[0].reset();
        fullReadMatrices[1].reset();
        fullReadMatrices128[0].reset();
        fullReadMatrices128[1].reset();
        strand = FORWARD_STRAND;
        matchWithSearches(read, maxED, occ);
        strand = REVERSE_C_STRAND;
        matchWithSearches(rc, maxED, occ);
        if (strat.metric == EDIT) return getUniqueTextOccurrences(occ, maxED);
        return getTextOccHamming(occ);
    }

    // ======================================================================================================
    // BEST (+x strata) mode: SearchStrategy::matchApproxBestPlusX (searchstrategy.cpp:714-746)
    // ======================================================================================================
    struct BestOcc {
        TextOcc t;            // range in concatenated-text coordinates
        len_t seqID = 0;      // assigned sequence
        len_t seqBegin = 0;   // begin inside it
        int found = -1;       // -1 not checked, 0 FOUND, 1 FOUND_WITH_TRIMMING, 2 NOT_FOUND
    };
    typedef std::vector<std::pair<bool, std::vector<BestOcc>>> OccVector;

    // fmindex.cpp:312-342 (+ InTextVerificationTask::doTask): one window, one zero in the first column
    void inTextVerificationOneString(len_t startPos, len_t endPos, len_t maxED, len_t minED, Occurrences& occ,
                                     const std::string& pattern) {
        // RUN_LENGTH_COMPRESSION: findSeqName calls checkTrimmedMatch instead (indexinterface.cpp:722-796, :870-888) — the same matrix
        // walk over the trimmed part of the occurrence's MATCHED STRING, which is text[startPos, endPos): the text the test attached
        // to the adapter stands in for it (the index itself holds none, bmove.cpp:590-596); no counters in that flavour.
        constexpr bool CNT = !RLC;
        if constexpr (RLC)
            if (!index.text) throw std::runtime_error("oracle: trimming on the b-move index needs the text beside it (orc_move_attach_text)");
        // use64Matrix(1, maxED) (fmindex.cpp:318; the RLC flavour: indexinterface.cpp:874-881): the 128-bit matrix beyond 10 errors
        if (!(BitParallelED64::LEFT >= 1 + maxED && BitParallelED64::MATRIX_MAX_ED >= maxED)) {
            oneStringOn<CNT>(fullReadMatrices128[strand], startPos, endPos, maxED, minED, occ, pattern);
            return;
        }
        oneStringOn<CNT>(fullReadMatrix(), startPos, endPos, maxED, minED, occ, pattern);
    }
    template <bool CNT, class MX>
    void oneStringOn(MX& matrix, len_t startPos, len_t endPos, len_t maxED, len_t minED, Occurrences& occ, const std::string& pattern) {
        Substring pat(pattern.data(), (len_t)pattern.size(), 0, (len_t)pattern.size(), FORWARD);
        if (!matrix.sequenceSet()) matrix.setSequence(pat);
        matrix.initializeMatrix(maxED, std::vector<uint32_t>(1, 0u));
        if (CNT) counters.inc(IN_TEXT_STARTED);
        Substring ref((const char*)index.text, index.textLength, startPos, endPos);
        const len_t size = ref.size();
        if (!matrix.inFinalColumn(size)) return;
        len_t i;
        for (i = 0; i < size; ++i) {
            if (CNT) counters.inc(MATRIX_ROWS);
            if (CNT) counters.inc(TEXT_BYTES);
            if (!matrix.computeRow(i + 1, ref.forwardAccessor(i))) break;
        }
        if (i <= size - matrix.getSizeOfFinalColumn()) {
            if (CNT) counters.inc(ABORTED_IN_TEXT_VERIF);
            return;
        }
        std::vector<len_t> refEnds;
        matrix.findClusterCenters(i, refEnds, maxED, minED);
        if (refEnds.empty()) {
            if (CNT) counters.inc(ABORTED_IN_TEXT_VERIF);
            return;
        }
        for (len_t refEnd : refEnds) {
            len_t bestScore = maxED + 1, bestBegin = 0;
            std::vector<std::pair<char, uint32_t>> cigar;
            matrix.traceBack(ref, refEnd, bestBegin, bestScore, &cigar);
            if (CNT) counters.inc(CIGARS_IN_TEXT_VERIFICATION);
            TextOcc t(Range(startPos + bestBegin, startPos + refEnd), bestScore, strand);
            t.cigar = cigar;
            occ.inTextOcc.emplace_back(std::move(t));
        }
    }
    // IndexInterface::generateCIGAR: findCIGAR of the occurrence's text range (bitparallelmatrix.h:460-527)
    void generateCIGAR(TextOcc& t, const std::string& seq) {
        if (strat.metric != EDIT || t.distance == 0) {
            t.cigar = {{'M', (uint32_t)seq.size()}};
            return;
        }
        // (RUN_LENGTH_COMPRESSION: the reference aligns the occurrence's matched string, indexinterface.h:966-971 — text[b, e) by
        // construction; here read from the text the test attached to the adapter)
        if constexpr (RLC)
            if (!index.text) throw std::runtime_error("oracle: CIGARs on the b-move index need the text beside it (orc_move_attach_text)");
        Substring ref((const char*)index.text, index.textLength, t.range.b, t.range.e);
        const Substring pat(seq.data(), (len_t)seq.size(), 0, (len_t)seq.size(), FORWARD);
        if (t.distance <= BitParallelED64::MATRIX_MAX_ED) { // (indexinterface.h:976-982)
            BitParallelED64 M;
            M.setSequence(pat);
            M.findCIGAR(ref, t.distance, t.cigar);
        } else {
            BitParallelED128 M;
            M.setSequence(pat);
            M.findCIGAR(ref, t.distance, t.cigar);
        }
    }
    // IndexInterface::findSeqName (indexinterface.cpp:799-899); returns 0 FOUND, 1 FOUND_WITH_TRIMMING, 2 NOT_FOUND
    int findSeqName(BestOcc& o, len_t largestStratum, const std::string& pattern) {
        const std::vector<len_t>& startPos = index.seqStarts;
        const len_t begin = o.t.range.b, end = o.t.range.e;
        len_t idx = (len_t)(std::upper_bound(startPos.begin(), startPos.end(), begin) - startPos.begin()) - 1;
        if (end <= startPos[idx + 1]) {
            o.seqID = idx;
            o.seqBegin = begin - startPos[idx];
            return 0;
        }
        if (strat.metric == HAMMING) return 2;
        Range range = o.t.range;
        if ((startPos[idx + 1] - begin) <= largestStratum) {
            idx++;
            if (idx + 1 >= startPos.size()) return 2; // (the reference would read past its vector here)
            range = Range(startPos[idx], std::min(end, startPos[idx + 1]));
        } else if ((end - startPos[idx + 1]) <= largestStratum) {
            range = Range(begin, startPos[idx + 1]);
        } else {
            return 2;
        }
        Occurrences occ;
        inTextVerificationOneString(range.b, range.e, largestStratum, 0, occ, pattern);
        if (occ.inTextOcc.empty()) return 2;
        o.t = *std::min_element(occ.inTextOcc.begin(), occ.inTextOcc.end());
        o.seqID = idx;
        o.seqBegin = o.t.range.b - startPos[idx];
        return 1;
    }
    // searchstrategy.h:490-523: one strand
    std::vector<TextOcc> mapRead(const std::string& read, len_t maxED, Strand st, len_t minD) {
        strand = st;
        if (maxED == 0) {
            std::vector<TextOcc> exact;
            exactMatchesOutput(read, exact);
            std::stable_sort(exact.begin(), exact.end());
            return exact;
        }
        Occurrences occ;
        matchWithSearches(read, maxED, occ);
        std::vector<TextOcc> v = strat.metric == EDIT ? getUniqueTextOccurrences(occ, maxED) : getTextOccHamming(occ);
        v.erase(std::remove_if(v.begin(), v.end(), [minD](const TextOcc& e) { return e.distance < minD; }), v.end());
        return v;
    }
    // searchstrategy.cpp:791-812
    bool processSeq(const std::string& seq, Strand st, len_t maxDist, OccVector& vec) {
        if (!vec[maxDist].first) {
            len_t minD = 0;
            while (minD < vec.size() && vec[minD].first) minD++;
            minD = std::min(minD, maxDist);
            for (auto& t : mapRead(seq, maxDist, st, minD)) {
                BestOcc b;
                b.t = t;
                vec[t.distance].second.emplace_back(std::move(b));
            }
            for (len_t i = minD; i <= maxDist; i++) vec[i].first = true;
        }
        for (len_t i = 0; i <= maxDist; i++)
            if (!vec[i].second.empty()) return true;
        return false;
    }
    // searchstrategy.cpp:536-571
    void checkAlignments(OccVector& ov, uint32_t& best, uint32_t l, uint32_t cutOff, const std::string& seq, Strand st) {
        strand = st;
        std::vector<BestOcc> trimmed, assigned;
        for (auto& o : ov[l].second) {
            if (!o.t.hasCigar()) generateCIGAR(o.t, seq);
            if (o.found < 0) o.found = findSeqName(o, cutOff, seq); // assignSequence (searchstrategy.h:1575-1593)
            if (o.found != 0) {
                if (o.found == 1 && o.t.distance > l) trimmed.emplace_back(std::move(o));
            } else {
                assigned.emplace_back(std::move(o));
                if (l < best) best = l;
            }
        }
        ov[l].second = std::move(assigned);
        for (auto& o : trimmed) {
            o.found = 0; // removeTrimmingLabel
            if (o.t.distance < ov.size()) ov[o.t.distance].second.emplace_back(std::move(o));
        }
    }
    // searchstrategy.cpp:623-712
    bool findBestAlignments(const std::string& read, const std::string& revC, OccVector& ovFW, OccVector& ovRC,
                            uint32_t x, uint32_t& best) {
        const len_t cutOff = (len_t)ovFW.size() - 1;
        best = cutOff + 1;
        bool bestFound = false;
        if (x == 0) {
            if (!ovFW[0].first) {
                strand = FORWARD_STRAND;
                std::vector<TextOcc> v;
                exactMatchesOutput(read, v);
                for (auto& t : v) {
                    BestOcc b;
                    b.t = t;
                    ovFW[0].second.emplace_back(b);
                }
                ovFW[0].first = true;
            }
            if (!ovRC[0].first) {
                strand = REVERSE_C_STRAND;
                std::vector<TextOcc> v;
                exactMatchesOutput(revC, v);
                for (auto& t : v) {
                    BestOcc b;
                    b.t = t;
                    ovRC[0].second.emplace_back(b);
                }
                ovRC[0].first = true;
            }
            if (!ovFW[0].second.empty() || !ovRC[0].second.empty()) {
                checkAlignments(ovFW, best, 0, cutOff, read, FORWARD_STRAND);
                checkAlignments(ovRC, best, 0, cutOff, revC, REVERSE_C_STRAND);
                if (best == 0) bestFound = true;
            }
        }
        uint32_t maxED = (best == 0) ? x : cutOff;
        uint32_t prevK = 0;
        auto hasUpdate = [&](const std::string& seq, Strand st, uint32_t k, OccVector& vec) {
            if (vec[k].first) return !vec[k].second.empty();
            return processSeq(seq, st, k, vec);
        };
        for (uint32_t k = std::max(x, (uint32_t)1); k <= maxED;) {
            bool update = false;
            update |= hasUpdate(read, FORWARD_STRAND, k, ovFW);
            update |= hasUpdate(revC, REVERSE_C_STRAND, k, ovRC);
            if (update) {
                for (len_t l = prevK + 1; l <= std::min(k, best + x); l++) {
                    checkAlignments(ovFW, best, l, maxED, read, FORWARD_STRAND);
                    checkAlignments(ovRC, best, l, maxED, revC, REVERSE_C_STRAND);
                }
            }
            if (bestFound) break;
            if (update && best < cutOff + 1) {
                bestFound = true;
                if (x == 0) break;
                prevK = k, k = std::min(best + x, maxED);
            } else {
                if (k == maxED) break;
                uint32_t step = (k < 5) ? 2 : 4;
                prevK = k;
                k = std::min(k + x + step, maxED);
            }
        }
        return bestFound;
    }
    // searchstrategy.cpp:714-746 (+ combineOccVectors :573-620, stable sort as DEVELOPER_MODE builds do); maxSupported =
    // getMaxSupportedDistanceForBestMapping of the strategy (capped by the caller where the device has no matrix)
    std::vector<BestOcc> matchApproxBestPlusX(const std::string& read, uint32_t x, uint32_t minIdentity, uint32_t maxSupported,
                                              uint32_t& best, uint32_t& nHits, bool& found) {
        fullReadMatrices[0].reset();
        fullReadMatrices[1].reset();
        fullReadMatrices128[0].reset();
        fullReadMatrices128[1].reset();
        noCIGAR = false;
        const std::string revC = revCompl(read);
        const len_t cutOff = std::min<len_t>(std::min<len_t>(13, maxSupported), ((len_t)read.size() * (100 - minIdentity)) / 100);
        OccVector ovFW(cutOff + 1), ovRC(cutOff + 1);
        found = findBestAlignments(read, revC, ovFW, ovRC, x, best);
        std::vector<BestOcc> matches;
        nHits = 0;
        if (!found) return matches;
        nHits = (uint32_t)(ovFW[best].second.size() + ovRC[best].second.size());
        auto compare = [](const BestOcc& a, const BestOcc& b) {
            return a.seqID < b.seqID || (a.seqID == b.seqID && a.seqBegin < b.seqBegin);
        };
        auto equal = [](const BestOcc& a, const BestOcc& b) { return a.seqID == b.seqID && a.seqBegin == b.seqBegin; };
        for (len_t i = best; i <= std::min<len_t>(best + x, cutOff); i++) {
            for (OccVector* ov : {&ovFW, &ovRC}) {
                auto& v = (*ov)[i].second;
                std::stable_sort(v.begin(), v.end(), compare);
                v.erase(std::unique(v.begin(), v.end(), equal), v.end());
                matches.insert(matches.end(), v.begin(), v.end());
            }
        }
        return matches;
    }

    // ======================================================================================================
    // SAM records of one read in ALL mode: matchApproxAllMap D) (searchstrategy.cpp:530-533) ->
    // generateOutputSingleEnd (:1824-1902) -> generateSE_SAM / generateSE_SAM_XATag (searchstrategy.h:1612-1641)
    // ======================================================================================================
    std::string samRecordsAll(const std::string& read, len_t maxED, const std::string& seqID, const std::string& qual,
                              const std::vector<std::string>& seqNames, bool unmappedSAM, bool xaTag) {
        const std::string revC = revCompl(read);
        std::string revQ = qual;
        std::reverse(revQ.begin(), revQ.end());
        noCIGAR = true;
        std::vector<TextOcc> result = matchApproxAll(read, maxED);
        std::vector<BestOcc> occs;
        for (auto& t : result) { // CIGARs (filterEditWithCIGARCalculation) and sequence assignment
            BestOcc o;
            o.t = t;
            strand = t.strand;
            const std::string& seq = t.strand == FORWARD_STRAND ? read : revC;
            generateCIGAR(o.t, seq);
            o.found = findSeqName(o, maxED, seq);
            if (o.found == 1 && !o.t.hasCigar()) generateCIGAR(o.t, seq);
            if (o.found != 2) occs.emplace_back(std::move(o));
        }
        if (occs.empty()) return unmappedSAM ? samUnmappedSE(seqID, read, qual) : std::string();
        auto minIt = std::min_element(occs.begin(), occs.end(),
                                      [](const BestOcc& a, const BestOcc& b) { return a.t.distance < b.t.distance; });
        const len_t minScore = minIt->t.distance;
        const len_t nHits = (len_t)std::count_if(occs.begin(), occs.end(), [&](const BestOcc& e) { return e.t.distance == minScore; });
        if (minIt != occs.begin()) std::iter_swap(occs.begin(), minIt);
        auto toSam = [&](const BestOcc& o) {
            SamOcc s2;
            s2.seqName = seqNames[o.seqID];
            s2.begin = o.seqBegin;
            s2.distance = o.t.distance;
            s2.revCompl = o.t.strand == REVERSE_C_STRAND;
            for (auto& p : o.t.cigar) s2.cigar += std::to_string(p.second) + p.first;
            return s2;
        };
        const bool rcFirst = occs[0].t.strand == REVERSE_C_STRAND;
        std::string out;
        if (xaTag) {
            std::vector<SamOcc> v;
            for (auto& o : occs) v.push_back(toSam(o));
            return samSingleEndXA(seqID, v, rcFirst ? revC : read, rcFirst ? revQ : qual, nHits);
        }
        out += samSingleEnd(seqID, toSam(occs[0]), rcFirst ? revC : read, rcFirst ? revQ : qual, nHits, minScore, true);
        for (size_t i = 1; i < occs.size(); i++) out += samSingleEnd(seqID, toSam(occs[i]), "*", "*", nHits, minScore, false);
        return out;
    }

    // nucleotide.h getRevComplWithN: complement ACGT, everything else N
    static std::string revCompl(const std::string& s) {
        std::string r(s.size(), 'N');
        for (size_t i = 0; i < s.size(); i++) {
            char c = s[s.size() - 1 - i], o = 'N';
            switch (c) {
            case 'A': o = 'T'; break;
            case 'C': o = 'G'; break;
            case 'G': o = 'C'; break;
            case 'T': o = 'A'; break;
            default: o = 'N';
            }
            r[i] = o;
        }
        return r;
    }
    // reads.h:43-58,97-101: upper-case, non-ACGT -> N
    static std::string cleanRead(const std::string& s) {
        std::string r = s;
        for (auto& c : r) {
            c = (char)toupper((unsigned char)c);
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T') c = 'N';
        }
        return r;
    }

    // ------------------------------------------------------------------ state
    Direction dir = BACKWARD;
    bool uniBackward = false;
    Strand strand = FORWARD_STRAND;
    std::vector<std::vector<FMPosExt>> stacks;
    std::vector<BitParallelED64> matrices;
    std::vector<BitParallelED128> matrices128; // the parts of a search whose upper bound exceeds 10 (indexinterface.cpp:391-398)
    std::vector<BitParallelED64N> matricesN;   // ORC_NARROW_BLOCKS=1: the narrow-block experiment (oracle_core.hpp)
    std::vector<BitParallelED32N> matricesN32; // ORC_NARROW_BLOCKS=32 (or "32inverted"): 32-bit words, 8-row blocks, up to 7 errors
    const bool narrowBlocks = getenv("ORC_NARROW_BLOCKS") != nullptr;
    const bool narrow32 = narrowBlocks && std::string(getenv("ORC_NARROW_BLOCKS")).rfind("32", 0) == 0;
    const bool narrowBroken = narrowBlocks && std::string(getenv("ORC_NARROW_BLOCKS")).find("inverted") != std::string::npos;
    BitParallelED64 fullReadMatrices[2];
    BitParallelED128 fullReadMatrices128[2]; // in-text verification beyond the 64-bit matrix (fmindex.h:240-246)

    void setDirection(Direction d, bool uni) { // indexinterface.h:771-779
        dir = d;
        uniBackward = uni;
    }
    // indexinterface.cpp:675-697
    void extendFMPos(const RangePair& parent, std::vector<FMPosExt>& stack, len_t row) {
        counters.inc(EXPANSIONS);
        for (len_t i = 1; i < 5; ++i) {
            RangePair child;
            bool ok = uniBackward ? index.extendBackwardUni(i, parent, child)
                                  : (dir == FORWARD ? index.extendForward(i, parent, child)
                                                    : index.extendBackward(i, parent, child));
            if (ok) {
                stack.emplace_back(IX::i2c((int)i), child, row + 1);
                counters.inc(NODE_COUNTER);
            }
        }
    }
    // indexinterface.cpp:1034-1049
    bool addChar(char c, RangePair& r) {
        int pos = IX::c2i(c);
        if (pos > -1) {
            counters.inc(EXPANSIONS);
            RangePair child;
            bool ok = uniBackward ? index.extendBackwardUni(pos, r, child)
                                  : (dir == FORWARD ? index.extendForward(pos, r, child)
                                                    : index.extendBackward(pos, r, child));
            r = child;
            if (ok) {
                counters.inc(NODE_COUNTER);
                return true;
            }
        }
        r = RangePair();
        return false;
    }
    // indexinterface.cpp:1016-1032
    RangePair matchStringBidirectionally(const Substring& pattern, RangePair r) {
        for (len_t i = 0; i < pattern.size(); i++) {
            if (!addChar(pattern[i], r)) break;
        }
        return r;
    }

    // === exact matching (indexinterface.cpp:918-1014) ===
    void exactMatchesOutput(const std::string& s, std::vector<TextOcc>& tOcc) {
        if (s.size() == 0) return;
        if constexpr (RLC) { // the RUN_LENGTH_COMPRESSION branches of :955-962, :976-981, :1002-1010: no in-text switch
            auto range = index.exactStart(); // SARangeBackwards with the toehold of the complete range
            for (len_t i = (len_t)s.size(); i-- > 0;) {
                int pos = IX::c2i(s[i]);
                if (pos == -1) return;
                counters.inc(EXPANSIONS);
                if (!index.extendExact(pos, range)) return;
                counters.inc(NODE_COUNTER);
            }
            for (len_t pos : index.beginPositionsExact(range, counters))
                tOcc.emplace_back(Range(pos, pos + (len_t)s.size()), 0, strand);
            counters.inc(TOTAL_REPORTED_POSITIONS, tOcc.size());
            return;
        } else {
        Range range = index.completeRange().sa;
        len_t i = (len_t)s.size();
        for (; i-- > 0;) {
            int pos = IX::c2i(s[i]);
            if (pos == -1) return;
            counters.inc(EXPANSIONS);
            if (!index.extendRangeBackward(pos, range, range)) return;
            counters.inc(NODE_COUNTER);
            if (range.width() <= index.switchPoint) break;
        }
        std::vector<len_t> p = index.getBeginPositions(range, 0, 0, counters);
        size_t before = tOcc.size();
        if (i != (len_t)-1) {
            // verifyInTextExact :918-943
            len_t charsMatched = (len_t)s.size() - i;
            len_t remaining = (len_t)s.size() - charsMatched;
            size_t added = 0;
            for (len_t pos : p) {
                len_t posInText = pos - remaining;
                bool ok = pos >= remaining;
                if (ok) {
                    // Substring(text,posInText,posInText+remaining).equals(prefix)
                    len_t endc = std::min<len_t>(posInText + remaining, index.textLength);
                    if (endc - posInText != remaining) ok = false;
                    for (len_t j = 0; ok && j < remaining; j++) {
                        counters.inc(TEXT_BYTES);
                        if ((char)index.text[posInText + j] != s[j]) ok = false;
                    }
                }
                if (ok) {
                    tOcc.emplace_back(Range(posInText, posInText + (len_t)s.size()), 0, strand);
                    added++;
                }
            }
            counters.inc(IN_TEXT_STARTED, p.size());
            // NB: the reference subtracts tOcc.size() (whole vector) :942
            counters.inc(ABORTED_IN_TEXT_VERIF, p.size() - tOcc.size());
            (void)added;
        } else {
            for (len_t pos : p)
                tOcc.emplace_back(Range(pos, pos + (len_t)s.size()), 0, strand);
        }
        (void)before;
        counters.inc(TOTAL_REPORTED_POSITIONS, tOcc.size());
        }
    }

    // === in-text verification (fmindex.cpp:245-310, indexhelpers.cpp:518-574) ===
    BitParallelED64& fullReadMatrix() { return fullReadMatrices[strand]; }

    void inTextVerification(const std::vector<len_t>& startPos, len_t maxED, len_t minED,
                            Occurrences& occ, const Substring& pattern, bool fixedStartPos) {
        len_t nZeros = fixedStartPos ? 1 : 2 * maxED + 1;
        // use64Matrix fmindex.h:240-246; else the 128-bit matrix (fmindex.cpp:276-283, :306-307)
        if (BitParallelED64::LEFT >= nZeros + maxED && BitParallelED64::MATRIX_MAX_ED >= maxED)
            inTextVerificationOn(fullReadMatrix(), startPos, maxED, minED, occ, pattern, nZeros);
        else
            inTextVerificationOn(fullReadMatrices128[strand], startPos, maxED, minED, occ, pattern, nZeros);
    }
    template <class Matrix>
    void inTextVerificationOn(Matrix& matrix, const std::vector<len_t>& startPos, len_t maxED, len_t minED,
                              Occurrences& occ, const Substring& pattern, len_t nZeros) {
        if constexpr (RLC) throw std::runtime_error("oracle: no in-text verification on the b-move index");
        else {
        if (!matrix.sequenceSet()) matrix.setSequence(pattern);
        matrix.initializeMatrix(maxED, std::vector<uint32_t>(nZeros, 0u));
        len_t nRows = matrix.getNumberOfRows();

        counters.inc(IN_TEXT_STARTED, startPos.size());
        for (len_t start : startPos) {
            len_t maxEnd = index.textLength - 1;
            len_t hEnd = std::min(maxEnd, start + nRows - 1);
            Substring ref((const char*)index.text, index.textLength, start, hEnd);
            len_t refBegin = ref.begin();
            len_t i;
            const len_t size = ref.size();
            if (!matrix.inFinalColumn(size)) continue;
            for (i = 0; i < size; ++i) {
                counters.inc(MATRIX_ROWS);
                counters.inc(TEXT_BYTES);
                if (!matrix.computeRow(i + 1, ref.forwardAccessor(i))) break;
            }
            if (i <= size - matrix.getSizeOfFinalColumn()) {
                counters.inc(ABORTED_IN_TEXT_VERIF);
                continue;
            }
            std::vector<len_t> refEnds;
            matrix.findClusterCenters(i, refEnds, maxED, minED);
            if (refEnds.empty()) {
                counters.inc(ABORTED_IN_TEXT_VERIF);
                continue;
            }
            for (len_t refEnd : refEnds) {
                len_t bestScore = maxED + 1, bestBegin = 0;
                std::vector<std::pair<char, uint32_t>> cigar;
                matrix.traceBack(ref, refEnd, bestBegin, bestScore, noCIGAR ? nullptr : &cigar);
                counters.inc(CIGARS_IN_TEXT_VERIFICATION);
                TextOcc t(Range(refBegin + bestBegin, refBegin + refEnd), bestScore, strand);
                if (!noCIGAR) t.cigar = cigar;
                occ.inTextOcc.emplace_back(std::move(t));
            }
        }
        }
    }

    // fmindex.cpp:245-265
    void verifyExactPartialMatchInText(const FMOcc& startMatch, len_t beginInPattern,
                                       len_t maxED, Occurrences& occ, len_t minED,
                                       const Substring& pattern) {
        if constexpr (!RLC) {
        counters.inc(IMMEDIATE_SWITCH);
        len_t startDiff = (beginInPattern == 0) ? 0 : beginInPattern + maxED;
        auto starts = index.getBeginPositions(startMatch.getRanges().sa, startDiff, 0, counters);
        inTextVerification(starts, maxED, minED, occ, pattern, beginInPattern == 0);
        }
    }

    // === edit-distance DFS (indexinterface.cpp:340-669, :1306-1325) ===
    template <class MX>
    void goToInTextVerificationEdit(const FMPosExt& node, const Search& s,
                                    const std::vector<Substring>& parts, Occurrences& occ,
                                    const Substring& pattern, len_t idx, MX* bpED,
                                    const FMOcc& sMatch, const std::vector<FMPosExt>& dOther,
                                    const std::vector<uint16_t>& iOther) {
        if constexpr (!RLC) { // (:345-348: the function does nothing in case of run-length compression)
        len_t st = parts[s.getLowestPartProcessedBefore(idx)].begin();
        len_t startDiff = st + s.getMaxED();
        if (st == 0) {
            startDiff = 0;
        } else if (dir == BACKWARD) {
            len_t row = node.getRow();
            len_t col = bpED->getFirstColumn(row);
            startDiff -= col + bpED->at(row, col);
        } else if (!dOther.empty()) {
            startDiff -= (len_t)(dOther.size() - iOther.size() + iOther.back());
        }
        auto pos = index.getBeginPositions(node.ranges.sa, startDiff, sMatch.shift, counters);
        inTextVerification(pos, s.getMaxED(), s.getMinED(), occ, pattern, st == 0);
        }
    }

    // The reference selects the matrix PER PART: 64-bit words where the search's upper bound at this part is at most 10, its 128-bit
    // matrix beyond (indexinterface.cpp:391-398); the function itself works on the IBitParallelED interface.
    void recApproxMatchEdit(const Search& s, const FMOcc& startMatch, Occurrences& occ,
                            const std::vector<Substring>& parts, int idx,
                            const std::vector<FMPosExt>& descPrevDir,
                            const std::vector<uint16_t>& initPrevDir,
                            const std::vector<FMPosExt>& descNotPrevDir,
                            const std::vector<uint16_t>& initNotPrevDir) {
        const size_t matrixIdx = s.getPart(idx) + (s.getDirection(idx) == BACKWARD) * s.getNumParts();
        // (the device's GeoN32: searches of batches up to 6 errors; ORC_NARROW32_MAX / ORC_NARROW32_SLACK override the two bounds for the
        // experiment that found them)
        const uint32_t n32Max = getenv("ORC_NARROW32_MAX") ? (uint32_t)atoi(getenv("ORC_NARROW32_MAX")) : 6u;
        const uint32_t n32Slack = getenv("ORC_NARROW32_SLACK") ? (uint32_t)atoi(getenv("ORC_NARROW32_SLACK")) : 2u;
        if (narrow32 && s.getMaxED() <= n32Max) { // (a phase it cannot hold falls back below)
            if (matricesN32.size() < matrices.size()) matricesN32.resize(matrices.size());
            // Wv = |initED| - 1 + maxED - initED.back() must not exceed DIAG (14): the phase otherwise runs on the reference's matrix, as the
            // device re-runs such a batch on GeoN (FLAG_CAPACITY)
            const bool dSw = s.getDirectionSwitch(idx);
            const std::vector<uint16_t>& ie = dSw ? initNotPrevDir : initPrevDir;
            uint32_t wv = s.getUpperBound(idx);
            if (!ie.empty()) {
                const uint16_t prevED = dSw ? *std::min_element(ie.begin(), ie.end()) : ie[0];
                wv = (uint32_t)ie.size() - 1 + s.getUpperBound(idx) - (ie.back() + (startMatch.distance - prevED));
            }
            if (wv + n32Slack <= BitParallelED32N::DIAG_R0) {
                g_narrow32Stats[0].fetch_add(1, std::memory_order_relaxed);
                matricesN32[matrixIdx].emulate = 64u | (narrowBroken ? 256u : 0u);
                recApproxMatchEditOn(&matricesN32[matrixIdx], s, startMatch, occ, parts, idx, descPrevDir, initPrevDir, descNotPrevDir, initNotPrevDir);
                return;
            }
            g_narrow32Stats[1].fetch_add(1, std::memory_order_relaxed);
            recApproxMatchEditOn(&matrices[matrixIdx], s, startMatch, occ, parts, idx, descPrevDir, initPrevDir, descNotPrevDir, initNotPrevDir);
        } else if (narrowBlocks && !narrow32) { // (experiment: ONE narrow-block matrix type in place of both, answering the predicate as the part's own would)
            if (matricesN.size() < matrices.size()) matricesN.resize(matrices.size());
            matricesN[matrixIdx].emulate = (s.getUpperBound(idx) <= BitParallelED64::MATRIX_MAX_ED ? 64u : 128u) | (narrowBroken ? 256u : 0u);
            recApproxMatchEditOn(&matricesN[matrixIdx], s, startMatch, occ, parts, idx, descPrevDir, initPrevDir, descNotPrevDir, initNotPrevDir);
        } else if (s.getUpperBound(idx) <= BitParallelED64::MATRIX_MAX_ED) {
            recApproxMatchEditOn(&matrices[matrixIdx], s, startMatch, occ, parts, idx, descPrevDir, initPrevDir, descNotPrevDir, initNotPrevDir);
        } else {
            if (matrices128.size() < matrices.size()) matrices128.resize(matrices.size());
            recApproxMatchEditOn(&matrices128[matrixIdx], s, startMatch, occ, parts, idx, descPrevDir, initPrevDir, descNotPrevDir, initNotPrevDir);
        }
    }
    template <class MX>
    void recApproxMatchEditOn(MX* bpED, const Search& s, const FMOcc& startMatch, Occurrences& occ,
                              const std::vector<Substring>& parts, int idx,
                              const std::vector<FMPosExt>& descPrevDir,
                              const std::vector<uint16_t>& initPrevDir,
                              const std::vector<FMPosExt>& descNotPrevDir,
                              const std::vector<uint16_t>& initNotPrevDir) {
        const Substring& p = parts[s.getPart(idx)];
        const len_t maxED = s.getUpperBound(idx);
        const Direction dirIdx = s.getDirection(idx);
        const bool dSwitch = s.getDirectionSwitch(idx);
        auto& stack = stacks[idx];

        const std::vector<uint16_t>& initEds = dSwitch ? initNotPrevDir : initPrevDir;
        const std::vector<FMPosExt>& descendants = dSwitch ? descNotPrevDir : descPrevDir;
        const std::vector<uint16_t>& initOther = dSwitch ? initPrevDir : initNotPrevDir;
        const std::vector<FMPosExt>& descOther = dSwitch ? descPrevDir : descNotPrevDir;

        setDirection(dirIdx, s.isUnidirectionalBackwards(idx));

        std::vector<uint32_t> initED;
        if (initEds.empty()) {
            initED = std::vector<uint32_t>(1, startMatch.distance);
        } else {
            uint16_t prevED = dSwitch ? *std::min_element(initEds.begin(), initEds.end())
                                      : initEds[0];
            uint32_t increase = startMatch.distance - prevED;
            initED.resize(initEds.size());
            for (size_t i = 0; i < initED.size(); i++) initED[i] = initEds[i] + increase;
        }
        if (!bpED->sequenceSet()) bpED->setSequence(p);
        bpED->initializeMatrix(maxED, initED);

        Cluster cluster(bpED->getSizeOfFinalColumn(), maxED, startMatch.getDepth(),
                        startMatch.shift);
        if (bpED->inFinalColumn(0)) {
            cluster.setValue(0, FMPosExt((char)0, startMatch.getRanges(), 0),
                             bpED->at(0, p.size()));
        }

        if (!descendants.empty()) {
            len_t maxRow = bpED->getNumberOfRows() - 1;
            for (len_t i = 0; i < descendants.size() && descendants[i].depth <= maxRow; i++) {
                std::vector<FMPosExt> remaining(descendants.begin() + i + 1, descendants.end());
                if (branchAndBound(bpED, cluster, descendants[i], s, idx, parts, occ, initOther,
                                   descOther, remaining)) {
                    return;
                }
            }
            if (descendants.back().depth == maxRow) return;
            RangePair pair = dSwitch ? startMatch.getRanges() : descendants.back().ranges;
            extendFMPos(pair, stack, descendants.back().depth);
        } else {
            extendFMPos(startMatch.getRanges(), stack, 0);
        }

        bool idxZero = idx == 0;
        Substring pattern(parts.back(), 0, parts.back().end(), FORWARD);
        len_t inTextSwitchPoint = index.switchPoint;
        static const std::vector<FMPosExt> noDesc;

        while (!stack.empty()) {
            const FMPosExt currentNode = stack.back();
            stack.pop_back();
            if (branchAndBound(bpED, cluster, currentNode, s, idx, parts, occ, initOther,
                               descOther, noDesc)) {
                continue;
            }
            if (!RLC && currentNode.ranges.width() <= inTextSwitchPoint && !idxZero) {
                goToInTextVerificationEdit(currentNode, s, parts, occ, pattern, idx, bpED,
                                           startMatch, descOther, initOther);
                continue;
            }
            extendFMPos(currentNode.ranges, stack, currentNode.depth);
        }
    }

    // indexinterface.cpp:529-561
    template <class MX>
    bool branchAndBound(MX* bpED, Cluster& cluster, const FMPosExt& currentNode,
                        const Search& s, len_t idx, const std::vector<Substring>& parts,
                        Occurrences& occ, const std::vector<uint16_t>& initOther,
                        const std::vector<FMPosExt>& descOther,
                        const std::vector<FMPosExt>& remainingDesc) {
        const len_t row = currentNode.depth;
        counters.inc(MATRIX_ROWS);
        bool validED = bpED->computeRow(row, currentNode.c);
        if (bpED->inFinalColumn(row)) {
            len_t clusterIdx = cluster.size() + row - bpED->getNumberOfRows();
            cluster.setValue(clusterIdx, currentNode, bpED->at(row, bpED->getNumberOfCols() - 1));
            if (!validED || bpED->onlyVerticalGapsLeft(row)) {
                goDeeper(cluster, idx + 1, s, parts, occ, descOther, initOther, remainingDesc);
                return true;
            }
        }
        return !validED;
    }

    // indexinterface.cpp:563-669
    void goDeeper(Cluster& cluster, len_t nIdx, const Search& s,
                  const std::vector<Substring>& parts, Occurrences& occ,
                  const std::vector<FMPosExt>& descOtherD,
                  const std::vector<uint16_t>& initOtherD, const std::vector<FMPosExt>& remDesc) {
        bool isEdge = s.isEdge(nIdx - 1);
        const len_t lowerBound = s.getLowerBound(nIdx - 1);
        if (isEdge) {
            if (nIdx == parts.size()) {
                auto matches = cluster.reportCentersAtEnd();
                for (auto& match : matches) {
                    if (match.isValid() && match.distance >= lowerBound) {
                        match.strand = strand;
                        occ.inFMOcc.emplace_back(match);
                    }
                }
            } else {
                FMOcc match = cluster.reportDeepestMinimum(this->dir);
                if (match.isValid() && match.distance >= lowerBound) {
                    Direction originalDir = this->dir;
                    recApproxMatchEdit(s, match, occ, parts, (int)nIdx, {}, {}, descOtherD,
                                       initOtherD);
                    setDirection(originalDir, s.isUnidirectionalBackwards(nIdx - 1));
                }
            }
            return;
        }
        std::vector<FMPosExt> descendants;
        std::vector<uint16_t> initEds;
        FMOcc newMatch = cluster.getClusterCentra((uint16_t)lowerBound, descendants, initEds);
        if (!newMatch.isValid()) return;
        descendants.insert(descendants.end(), remDesc.begin(), remDesc.end());
        for (len_t i = 0; i < descendants.size(); i++) descendants[i].depth = i + 1;
        len_t maxEDNext = s.getUpperBound(nIdx);
        while (initEds.back() > maxEDNext) initEds.pop_back();
        bool switchAfter = s.getDirectionSwitch(nIdx);
        if (switchAfter) {
            if (!descendants.empty()) {
                newMatch.pos.ranges = descendants.back().ranges;
                newMatch.distance = *std::min_element(initEds.begin(), initEds.end());
            }
            Direction originalDir = this->dir;
            recApproxMatchEdit(s, newMatch, occ, parts, (int)nIdx, descendants, initEds,
                               descOtherD, initOtherD);
            setDirection(originalDir, s.isUnidirectionalBackwards(nIdx - 1));
        } else {
            recApproxMatchEdit(s, newMatch, occ, parts, (int)nIdx, descendants, initEds,
                               descOtherD, initOtherD);
        }
    }

    // indexinterface.cpp:1306-1325
    void recApproxMatchEditEntry(const Search& search, const FMOcc& startMatch, Occurrences& occ,
                                 const std::vector<Substring>& parts, int idx) {
        if (RLC || startMatch.getRanges().width() > index.switchPoint) {
            counters.inc(SEARCH_STARTED);
            recApproxMatchEdit(search, startMatch, occ, parts, idx, {}, {}, {}, {});
            return;
        }
        verifyExactPartialMatchInText(
            startMatch, parts[search.getLowestPartProcessedBefore(idx)].begin(), search.getMaxED(),
            occ, search.getMinED(), Substring(parts.back(), 0, parts.back().end(), FORWARD));
    }

    // === Hamming DFS (indexinterface.cpp:1211-1304; fmindex.cpp:344-428) ===
    void inTextVerificationHammingRange(const SARange& r, const Substring& pattern, len_t maxEDFull,
                                        len_t minEDFull, len_t lengthBefore, Occurrences& occ) {
        if constexpr (!RLC) {
        const len_t pSize = pattern.size();
        for (len_t i = r.b; i < r.e; i++) {
            len_t Tb = index.findSA(i, counters);
            counters.inc(IN_TEXT_STARTED);
            Tb = (Tb > lengthBefore) ? Tb - lengthBefore : 0;
            len_t Te = Tb + pSize;
            if (Te > index.textLength) continue;
            len_t score = 0;
            for (len_t j = 0; j < pSize; j++) {
                counters.inc(TEXT_BYTES);
                score = score + ((char)index.text[Tb + j] != pattern.forwardAccessor(j));
                if (score > maxEDFull) break;
            }
            if (score <= maxEDFull && score >= minEDFull)
                occ.inTextOcc.emplace_back(Range(Tb, Te), score, strand);
        }
        }
    }
    void recApproxMatchHamming(const Search& s, const FMOcc& startMatch, Occurrences& occ,
                               const std::vector<Substring>& parts, int idx) {
        const Substring& p = parts[s.getPart(idx)];
        const len_t pSize = p.size();
        const Direction d = s.getDirection(idx);
        const len_t maxED = s.getUpperBound(idx);
        const len_t minED = s.getLowerBound(idx);
        setDirection(d, s.isUnidirectionalBackwards(idx));
        std::vector<len_t> vec(p.size() + 1, 0);
        vec[0] = startMatch.distance;
        auto& stack = stacks[idx];
        extendFMPos(startMatch.getRanges(), stack, 0);
        while (!stack.empty()) {
            const FMPosExt node = stack.back();
            stack.pop_back();
            if (!RLC && node.ranges.width() <= index.switchPoint) { // (:1245-1249: FM flavour only)
                // fmindex.cpp:409-428
                len_t lengthBefore =
                    ((idx == 0) ? 0 : parts[s.getLowestPartProcessedBefore(idx)].begin()) -
                    (dir == BACKWARD) * (node.depth);
                len_t fullSize = parts.back().end();
                Substring pattern(parts[0], 0, fullSize, FORWARD);
                inTextVerificationHammingRange(node.ranges.sa, pattern, s.getMaxED(), s.getMinED(),
                                               lengthBefore, occ);
                continue;
            }
            len_t row = node.getRow();
            vec[row] = vec[row - 1] + (node.c != p[row - 1]);
            if (vec[row] > maxED) continue;
            if (row == pSize) {
                if (vec[row] >= minED) {
                    FMOcc match(node.ranges, vec[row], startMatch.getDepth() + pSize, strand);
                    if (s.isEnd(idx)) {
                        occ.inFMOcc.emplace_back(match);
                    } else {
                        recApproxMatchHamming(s, match, occ, parts, idx + 1);
                        setDirection(s.getDirection(idx), s.isUnidirectionalBackwards(idx));
                    }
                }
                continue;
            }
            extendFMPos(node.ranges, stack, node.depth);
        }
    }

    // === naive backtracking (indexinterface.cpp:1055-1209) ===
    void approxMatchesNaive(const std::string& pattern, len_t maxED, Occurrences& occurrences) {
        matrices.assign(2, BitParallelED64());
        matrices128.assign(2, BitParallelED128());
        if (maxED > BitParallelED64::MATRIX_MAX_ED) approxMatchesNaiveOn(&matrices128.front(), pattern, maxED, occurrences); // (:1063)
        else approxMatchesNaiveOn(&matrices.front(), pattern, maxED, occurrences);
    }
    template <class MX>
    void approxMatchesNaiveOn(MX* matrix, const std::string& pattern, len_t maxED, Occurrences& occurrences) {
        setDirection(BACKWARD, true);
        Substring p(pattern.data(), (len_t)pattern.size(), 0, (len_t)pattern.size(), BACKWARD);
        matrix->setSequence(p);
        matrix->initializeMatrix(maxED);
        std::vector<FMPosExt> stack;
        extendFMPos(index.completeRange(), stack, 0);
        len_t lastCol = (len_t)pattern.size();
        Substring fwd(pattern.data(), (len_t)pattern.size(), 0, (len_t)pattern.size(), FORWARD);
        while (!stack.empty()) {
            const FMPosExt currentNode = stack.back();
            stack.pop_back();
            len_t row = currentNode.depth;
            if (row >= matrix->getNumberOfRows()) continue;
            counters.inc(MATRIX_ROWS);
            bool valid = matrix->computeRow(row, currentNode.c);
            if (!valid) continue;
            if (matrix->inFinalColumn(row)) {
                if (matrix->at(row, lastCol) <= maxED)
                    occurrences.inFMOcc.emplace_back(currentNode.ranges, matrix->at(row, lastCol),
                                                     currentNode.depth, strand);
            }
            if constexpr (!RLC) { // (:1120-1132)
            if (currentNode.ranges.width() <= index.switchPoint) {
                auto startPos = index.getBeginPositions(currentNode.ranges.sa, 0, 0, counters);
                inTextVerification(startPos, maxED, 0, occurrences, fwd, true);
                continue;
            }
            }
            extendFMPos(currentNode.ranges, stack, currentNode.depth);
        }
    }
    void approxMatchesNaiveHamming(const std::string& pattern, len_t maxED, Occurrences& occ) {
        setDirection(BACKWARD, true);
        std::vector<len_t> vec(pattern.size() + 1, 0);
        std::vector<FMPosExt> stack;
        extendFMPos(index.completeRange(), stack, 0);
        Substring fwd(pattern.data(), (len_t)pattern.size(), 0, (len_t)pattern.size(), FORWARD);
        while (!stack.empty()) {
            const FMPosExt node = stack.back();
            stack.pop_back();
            if (!RLC && node.ranges.width() <= index.switchPoint) { // (:1170-1175)
                inTextVerificationHammingRange(node.ranges.sa, fwd, maxED, 0, 0, occ);
                continue;
            }
            len_t row = node.getRow();
            vec[row] = vec[row - 1] + (node.c != pattern[pattern.size() - row]);
            if (vec[row] > maxED) continue;
            if (row == pattern.size()) {
                occ.inFMOcc.emplace_back(node.ranges, vec[row], node.depth, strand);
                continue;
            }
            extendFMPos(node.ranges, stack, node.depth);
        }
    }

    // === partitioning (searchstrategy.cpp:141-419) ===
    void calculateExactMatchRanges(const std::string& pattern, std::vector<Substring>& parts,
                                   std::vector<RangePair>& exactMatchRanges) {
        setDirection(FORWARD, false);
        len_t wordSize = index.wordSize;
        for (len_t i = 0; i < parts.size(); ++i) {
            auto& current = parts[i];
            len_t size = current.size();
            len_t start = current.begin() + ((size >= wordSize) ? wordSize : 0);
            RangePair initRanges =
                (size >= wordSize)
                    ? index.lookUpInKmerTable(pattern.data(), current.begin(), start)
                    : index.completeRange();
            exactMatchRanges[i] =
                matchStringBidirectionally(Substring(current, start, current.end()), initRanges);
        }
        setDirection(BACKWARD, true);
        auto& lastPart = parts.back();
        lastPart.setDirection(BACKWARD);
        len_t size = lastPart.size();
        len_t end = (size >= wordSize) ? lastPart.end() - wordSize : lastPart.end();
        RangePair initRanges = (size >= wordSize)
                                   ? index.lookUpInKmerTable(pattern.data(), end, lastPart.end())
                                   : index.completeRange();
        exactMatchRanges.back() =
            matchStringBidirectionally(Substring(lastPart, lastPart.begin(), end), initRanges);
    }

    Substring mkPart(const std::string& pattern, len_t b, len_t e) {
        return Substring(pattern.data(), (len_t)pattern.size(), b, e, FORWARD);
    }

    void partitionUniform(const std::string& pattern, std::vector<Substring>& parts, int numParts,
                          std::vector<RangePair>& exactMatchRanges) {
        for (int i = 0; i < numParts; i++) {
            parts.push_back(mkPart(pattern, (len_t)((i * 1.0 / numParts) * pattern.size()),
                                   (len_t)(((i + 1) * 1.0 / numParts) * pattern.size())));
        }
        parts.back().setEnd((len_t)pattern.size());
        calculateExactMatchRanges(pattern, parts, exactMatchRanges);
    }
    void partitionStatic(const std::string& pattern, std::vector<Substring>& parts, int numParts,
                         int maxScore, std::vector<RangePair>& exactMatchRanges) {
        std::vector<double> begins = strat.getBegins(numParts, maxScore);
        int pSize = (int)pattern.size();
        parts.push_back(mkPart(pattern, 0, (len_t)(begins[0] * pSize)));
        for (unsigned i = 0; i < begins.size() - 1; i++)
            parts.push_back(mkPart(pattern, (len_t)(begins[i] * pSize), (len_t)(begins[i + 1] * pSize)));
        parts.push_back(mkPart(pattern, (len_t)(begins.back() * pSize), (len_t)pattern.size()));
        calculateExactMatchRanges(pattern, parts, exactMatchRanges);
    }
    // searchstrategy.cpp:381-419
    int seed(const std::string& pattern, std::vector<Substring>& parts, int numParts, int maxScore,
             std::vector<RangePair>& exactMatchRanges) {
        size_t pSize = pattern.size();
        bool useKmerTable =
            ((size_t)numParts * index.wordSize < (pSize * 2) / 3) && (pSize >= strat.useKmerCutOff);
        int wSize = useKmerTable ? (int)index.wordSize : 1;
        std::vector<double> seedPercent = strat.getSeedingPositions(numParts, maxScore);
        std::vector<int> seeds;
        seeds.emplace_back(0);
        for (int i = 1; i < numParts - 1; i++)
            seeds.emplace_back((int)((seedPercent[i - 1] * pSize) - (wSize / 2)));
        for (int i = 0; i < numParts - 1; i++)
            parts.push_back(mkPart(pattern, (len_t)seeds[i], (len_t)(seeds[i] + wSize)));
        parts.push_back(mkPart(pattern, (len_t)(pSize - wSize), (len_t)pSize));
        exactMatchRanges.resize(numParts);
        for (int i = 0; i < numParts; i++) {
            exactMatchRanges[i] =
                useKmerTable
                    ? index.lookUpInKmerTable(pattern.data(), parts[i].begin(), parts[i].end())
                    : index.rangeOfSingleChar(parts[i][0]);
        }
        return numParts * wSize;
    }
    // searchstrategy.cpp:299-379
    void partitionDynamic(const std::string& pattern, std::vector<Substring>& parts, int numParts,
                          int maxScore, std::vector<RangePair>& exactMatchRanges) {
        int matchedChars = seed(pattern, parts, numParts, maxScore, exactMatchRanges);
        len_t pSize = (len_t)pattern.size();
        std::vector<uint64_t> weights = strat.getWeights(numParts, maxScore);
        Direction d = FORWARD;
        int partToExtend = 0;
        // matchedChars == 0 only if wordSize == 0 (seedIfNoKmers :255): not
        // reachable with wordSize >= 1
        for (len_t j = (len_t)matchedChars; j < pSize; j++) {
            uint64_t maxRangeWeighted = 0;
            for (int i = 0; i < numParts; i++) {
                bool noLeftExtension = (i == 0) || parts[i].begin() == parts[i - 1].end();
                bool noRightExtension =
                    (i == numParts - 1) || parts[i].end() == parts[i + 1].begin();
                if (noLeftExtension && noRightExtension) continue;
                if (exactMatchRanges[i].width() * weights[i] > maxRangeWeighted) {
                    maxRangeWeighted = exactMatchRanges[i].width() * weights[i];
                    partToExtend = i;
                    if (noLeftExtension) d = FORWARD;
                    else if (noRightExtension) d = BACKWARD;
                    else
                        d = (exactMatchRanges[i - 1].width() < exactMatchRanges[i + 1].width())
                                ? BACKWARD
                                : FORWARD;
                }
            }
            if (maxRangeWeighted == 0) {
                // extendParts :283-297
                for (len_t i = 0; i < parts.size(); i++) {
                    if ((i != parts.size() - 1) && (parts[i].end() != parts[i + 1].begin()))
                        parts[i].setEnd(parts[i + 1].begin());
                    if ((i != 0) && (parts[i].begin() != parts[i - 1].end()))
                        parts[i].setBegin(parts[i - 1].end());
                }
                return;
            }
            char c;
            if (d == FORWARD) {
                parts[partToExtend].setEnd(parts[partToExtend].end() + 1);
                c = pattern[parts[partToExtend].end() - 1];
            } else {
                parts[partToExtend].setBegin(parts[partToExtend].begin() - 1);
                c = pattern[parts[partToExtend].begin()];
            }
            setDirection(d, partToExtend == numParts - 1);
            addChar(c, exactMatchRanges.at(partToExtend));
        }
    }

    // === per-strand driver (searchstrategy.cpp:425-493, :1181-1254) ===
    void matchWithSearches(const std::string& seq, len_t k, Occurrences& occs,
                           len_t minDistance = 0) {
        if (!strat.supports(k)) throw std::runtime_error("oracle: distance not supported by strategy");
        len_t numParts = strat.calculateNumParts(k);
        std::vector<RangePair> exactMatchRanges(numParts);
        std::vector<Substring> parts;
        // partition() :141-156
        if (!(numParts >= (len_t)seq.size() || numParts == 1)) {
            if (strat.partition == UNIFORM) partitionUniform(seq, parts, numParts, exactMatchRanges);
            else if (strat.partition == STATIC) partitionStatic(seq, parts, numParts, k, exactMatchRanges);
            else partitionDynamic(seq, parts, numParts, k, exactMatchRanges);
        }
        if (parts.empty()) {
            Occurrences local;
            std::vector<TextOcc> textOccs;
            if (strat.metric == EDIT) {
                approxMatchesNaive(seq, k, local);
                textOccs = getUniqueTextOccurrences(local, k);
            } else {
                approxMatchesNaiveHamming(seq, k, local);
                textOccs = getTextOccHamming(local);
            }
            for (auto& t : textOccs) occs.inTextOcc.emplace_back(std::move(t));
            return;
        }
        for (uint16_t i = 0; !RLC && i < numParts; i++) { // (#ifndef RUN_LENGTH_COMPRESSION, searchstrategy.cpp:461)
            size_t width = exactMatchRanges[i].width();
            if (width != 0 && width <= index.switchPoint) {
                const auto& part = parts[i];
                FMOcc startMatch(exactMatchRanges[i], 0, part.size());
                if (strat.metric == EDIT) {
                    Substring pattern(parts.back(), 0, parts.back().end(), FORWARD);
                    verifyExactPartialMatchInText(startMatch, part.begin(), k, occs, minDistance,
                                                  pattern);
                } else {
                    // fmindex.cpp:344-356
                    len_t pSize = parts.back().end();
                    Substring pattern(parts[0], 0, pSize, FORWARD);
                    inTextVerificationHammingRange(startMatch.getRanges().sa, pattern, k,
                                                   minDistance, part.begin(), occs);
                }
            }
        }
        const std::vector<Search>& searches = strat.createSearches(k, exactMatchRanges);
        stacks.assign(numParts, {});
        matrices.assign(2 * parts.size(), BitParallelED64());
        matrices128.assign(k > BitParallelED64::MATRIX_MAX_ED ? 2 * parts.size() : 0, BitParallelED128());
        matricesN.assign(narrowBlocks && !narrow32 ? 2 * parts.size() : 0, BitParallelED64N());
        matricesN32.assign(narrow32 ? 2 * parts.size() : 0, BitParallelED32N());
        for (const Search& s : searches) doRecSearch(s, parts, occs, exactMatchRanges);
    }

    void startIdx(const Search& s, const FMOcc& startMatch, Occurrences& occ,
                  const std::vector<Substring>& parts, int idx) {
        if (strat.metric == EDIT) recApproxMatchEditEntry(s, startMatch, occ, parts, idx);
        else recApproxMatchHamming(s, startMatch, occ, parts, idx);
    }

    void doRecSearch(const Search& s, std::vector<Substring>& parts, Occurrences& occ,
                     const std::vector<RangePair>& exactMatchRanges) {
        if (s.getUpperBound(0) > 0) {
            s.setDirectionsInParts(parts);
            FMOcc startMatch(index.completeRange(), 0, 0);
            startIdx(s, startMatch, occ, parts, 0);
            return;
        }
        int first = (int)s.getPart(0);
        RangePair startRange = exactMatchRanges[first];
        if (startRange.width() > index.switchPoint) {
            s.setDirectionsInParts(parts);
            uint16_t partInSearch = 1;
            len_t exactLength = parts[first].size();
            while (s.getUpperBound(partInSearch) == 0) {
                setDirection(s.getDirection(partInSearch), s.isUnidirectionalBackwards(partInSearch));
                const auto& part = parts[s.getPart(partInSearch)];
                startRange = matchStringBidirectionally(part, startRange);
                if (startRange.empty()) return;
                exactLength += part.size();
                partInSearch++;
            }
            FMOcc startMatch(startRange, 0, exactLength);
            startIdx(s, startMatch, occ, parts, partInSearch);
        }
    }

    // === post-processing (indexinterface.cpp:1331-1491) ===
    std::vector<TextOcc> getTextOccHamming(Occurrences& occ) {
        counters.inc(TOTAL_REPORTED_POSITIONS, occ.inTextOcc.size());
        occ.eraseDoublesFM();
        len_t size = occ.inFMOcc.empty() ? 0 : occ.inFMOcc[0].getDepth();
        for (size_t fi = 0; fi < occ.inFMOcc.size(); fi++) {
            const auto& f = occ.inFMOcc[fi];
            const SARange& saRange = f.getRanges().sa;
            counters.inc(TOTAL_REPORTED_POSITIONS, saRange.width());
            const uint64_t lf0 = counters.c[LF_STEPS];
            for (len_t p : index.textPositions(f.getRanges(), counters)) // getTextPositionsFromSARange
                occ.inTextOcc.emplace_back(Range(p, p + size), f.distance, f.strand);
            noteSurvivingDuplicate(occ, fi, lf0);
        }
        occ.eraseDoublesAndSortText();
        return std::move(occ.inTextOcc);
    }

    // (not in the reference: accounting of the repeated work described at SURVIVING_DUP_ROWS, oracle_core.hpp)
    void noteSurvivingDuplicate(const Occurrences& occ, size_t fi, uint64_t lfBefore) {
        for (size_t j = 0; j < fi; j++)
            if (occ.inFMOcc[j] == occ.inFMOcc[fi]) {
                counters.inc(SURVIVING_DUP_ROWS, occ.inFMOcc[fi].getRanges().sa.width());
                counters.inc(SURVIVING_DUP_LF, counters.c[LF_STEPS] - lfBefore);
                return;
            }
    }

    std::vector<TextOcc> getUniqueTextOccurrences(Occurrences& occ, len_t maxED) {
        counters.inc(TOTAL_REPORTED_POSITIONS, occ.inTextOcc.size());
        occ.eraseDoublesFM();
        for (size_t fi = 0; fi < occ.inFMOcc.size(); fi++) {
            const auto& f = occ.inFMOcc[fi];
            const SARange& saRange = f.getRanges().sa;
            counters.inc(TOTAL_REPORTED_POSITIONS, saRange.width());
            len_t depth = f.getDepth(), distance = f.distance, shift = f.shift;
            const uint64_t lf0 = counters.c[LF_STEPS];
            for (len_t p : index.textPositions(f.getRanges(), counters)) { // getTextPositionsFromSARange
                len_t startPos = p + shift;
                occ.inTextOcc.emplace_back(Range(startPos, startPos + depth), distance, f.strand);
            }
            noteSurvivingDuplicate(occ, fi, lf0);
        }
        occ.eraseDoublesAndSortText();
        std::vector<TextOcc> nonRedundantOcc;
        len_t maxDiff = 2 * maxED;
        len_t prevBegin = std::numeric_limits<len_t>::max();
        len_t prevDepth = std::numeric_limits<len_t>::max();
        len_t prevED = maxED + 1;
        for (auto& o : occ.inTextOcc) {
            len_t ob = o.range.b;
            len_t diff = ob > prevBegin ? ob - prevBegin : prevBegin - ob;
            if (diff == 0) continue;
            if (diff <= maxDiff) {
                if (o.distance > prevED || (o.distance == prevED && o.range.width() >= prevDepth))
                    continue;
                nonRedundantOcc.pop_back();
            }
            prevBegin = o.range.b;
            prevED = o.distance;
            prevDepth = o.range.width();
            nonRedundantOcc.emplace_back(std::move(o));
        }
        return nonRedundantOcc;
    }
};

typedef MatcherT<Index> Matcher;

} // namespace orc

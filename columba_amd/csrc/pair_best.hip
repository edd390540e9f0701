// Pairing of read pairs in BEST (+x strata) mode (include/columba_amd.h, section "paired-end reads in BEST mode"): host code, a
// translation unit of its own.  Reference: SearchStrategy::matchApproxPairedEndBestPlusX (src/searchstrategy.cpp:1091-1179) over
// processCombFF / RF / FR (:936-1062), processComb (:834-912), processSeq (:778-812), pairOccurrencesForBestMapping (:1743-1815),
// handleTrimmedOccs (:814-832), mergeOrMovePairs (:914-934); without a concordant pair pairDiscordantlyBest (:1664-1741) with
// mapStratum (searchstrategy.h:1354-1361), addDiscPairs (:1518-1585), findBestMapping (:1648-1662) -> findBestAlignments (:623-712)
// -> checkAlignments (:536-567) / combineOccVectors (:569-621), addUnpairedMatches (:1401-1461), addOneUnmapped (:1463-1516),
// addBothUnmapped (searchstrategy.h:1236-1247); the records by generateSAMPairedEnd (:1904-1970) in the order
// OutputWriter::writeChunks prints them (fastq.cpp:662-702).
//
// The reference walks ONE pair through its strata and calls mapRead (searchstrategy.h:490-519: the ALL-mode search of one strand of
// one mate at one distance, filtered, occurrences below minD dropped) whenever it needs a stratum it has not looked at.  Which
// strata that are depends on what the earlier ones held, so a chunk of pairs cannot be planned ahead.  Here the walk of a pair is a
// function of the mapRead results it has been given: it runs until it needs a list it does not hold, reports that request
// (cmb_pair_best_advance) and is run again from the start once the list has arrived (cmb_pair_best_supply) — every run makes the
// same decisions on the same data, so it gets further each time; the caller turns the requests of a whole chunk into a few device
// batches (one per distance).  A list is the device's ALL-mode result of that read at that distance with each strand filtered by
// itself (cmb_batch_filter_per_strand) and the alignments of cmb_batch_alignments.
#include "../../include/columba_amd.h"
#include "host_sam.hpp"

#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <tuple>
#include <vector>

namespace cmb {
int failWith(int code, const std::string& msg); // columba_amd.hip
}
using namespace cmb;

namespace {
enum { UNCHECKED = 0, FOUND = 1, FOUND_WITH_TRIMMING = 2, NOT_FOUND = 3 }; // SeqNameFound + "not looked at yet"

struct BOcc { // a TextOcc of the reference, as far as the pairing reads it
    uint32_t indexBegin = 0, width = 0, distance = 0; // in the concatenated text
    uint32_t seqId = 0, seqBegin = 0;                 // what findSeqName assigns (from the device for occurrences inside one sequence)
    uint8_t strand = 0, second = 0, state = UNCHECKED, spans = 0;
    std::vector<uint16_t> ops;
    bool assigned() const { return state == FOUND || state == FOUND_WITH_TRIMMING; }
    uint32_t rangeBegin() const { return assigned() ? seqBegin : indexBegin; } // the range turns relative once a sequence is assigned
    uint32_t indexEnd() const { return indexBegin + width; }
};
bool occLess(const BOcc& a, const BOcc& b) { // TextOcc::operator< (indexhelpers.h:779-795)
    if (a.rangeBegin() != b.rangeBegin()) return a.rangeBegin() < b.rangeBegin();
    if (a.distance != b.distance) return a.distance < b.distance;
    return a.width < b.width;
}
struct Stratum {
    bool done = false;
    std::vector<BOcc> v;
};
typedef std::vector<Stratum> OccVector; // one entry per distance 0 .. cut-off

struct BPair { // PairedTextOccs
    BOcc up, down;
    bool upValid = true, downValid = true;
    uint32_t fragSize = 0, distance = 0;
    bool discordant = false;
    std::string upLine, downLine;
};

struct Need {
    uint32_t mate, strand, k;
};

struct Unpaired { // an unpaired record, kept as data until the sequence names are at hand (cmb_pair_best_sam)
    BOcc o;
    uint32_t bestCount, best;
    bool first, unmapped;
    uint32_t mate;
};

typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t> TrimKey; // mate, strand, begin, width, distance, stratum

struct PairState {
    std::string id[2], seq[2], rc[2], qual[2], rqual[2];
    uint32_t cutOff[2] = {0, 0};
    std::map<uint32_t, std::vector<BOcc>> lists[2][2]; // [mate][strand][distance]: mapRead's result, as supplied
    std::map<TrimKey, std::pair<bool, BOcc>> trims;     // findSeqName's outcome for occurrences that run over the end of a sequence
    bool seeded = false, read2done = false; // addSingleEndedForBest: the walk starts from single-end results
    std::vector<BOcc> seeds[2];
    bool finished = false;
    std::vector<BPair> pairs;       // the outcome: pairs ...
    std::vector<Unpaired> unpaired; // ... and unpaired records
};
} // namespace

struct cmb_pair_best {
    cmb_pair_params prm;
    uint32_t x = 0, minIdentity = 0, maxSupported = 0;
    int metric = CMB_METRIC_EDIT;
    cmb_index* textIndex = nullptr;
    cmb_pair_trim_fn trimFn = nullptr;
    void* trimUser = nullptr;
    std::vector<PairState> pairs;
};

namespace {
// one walk of one pair through the reference's logic; throws Need at the first list it does not hold
struct Walk {
    cmb_pair_best& B;
    uint32_t pairIndex;
    PairState& P;
    OccVector ov[2][2]; // [mate][strand]
    std::vector<BPair> pairs;

    Walk(cmb_pair_best& b, uint32_t i) : B(b), pairIndex(i), P(b.pairs[i]) {
        for (int m = 0; m < 2; m++)
            for (int s = 0; s < 2; s++) ov[m][s].assign(P.cutOff[m] + 1, Stratum());
        if (P.seeded) { // addSingleEndedForBest (:1064-1089): the single-end matches in their strata; read 1 counts as looked at everywhere
            for (int m = 0; m < 2; m++)
                for (const BOcc& o : P.seeds[m]) ov[m][o.strand][o.distance].v.push_back(o);
            for (int s = 0; s < 2; s++) {
                for (Stratum& st : ov[0][s]) st.done = true;
                for (Stratum& st : ov[1][s]) st.done = P.read2done;
            }
        }
    }
    const std::string& seqOf(int m, int s) const { return s ? P.rc[m] : P.seq[m]; }

    // mapRead (searchstrategy.h:490-519)
    std::vector<BOcc> mapRead(uint32_t m, uint32_t s, uint32_t maxED, uint32_t minD) {
        auto it = P.lists[m][s].find(maxED);
        if (it == P.lists[m][s].end()) throw Need{m, s, maxED};
        std::vector<BOcc> r;
        for (const BOcc& o : it->second)
            if (o.distance >= minD) r.push_back(o);
        return r;
    }
    // processSeq (:778-812)
    bool processSeq(uint32_t m, uint32_t s, uint32_t maxDist) {
        OccVector& v = ov[m][s];
        if (!v[maxDist].done) {
            uint32_t minD = 0;
            while (minD < v.size() && v[minD].done) minD++;
            minD = std::min(minD, maxDist);
            for (BOcc& o : mapRead(m, s, maxDist, minD)) {
                const uint32_t d = o.distance;
                if (d < v.size()) v[d].v.push_back(std::move(o));
            }
            for (uint32_t i = minD; i <= maxDist; i++) v[i].done = true;
        }
        for (uint32_t i = 0; i <= maxDist; i++)
            if (!v[i].v.empty()) return true;
        return false;
    }
    // assignSequence (searchstrategy.h:1579-1599) over findSeqName (indexinterface.cpp:799-899): the outcome sticks to the occurrence
    int assign(BOcc& o, uint32_t maxED) {
        if (o.state != UNCHECKED) return o.state;
        if (!o.spans) return o.state = FOUND;
        if (B.metric == CMB_METRIC_HAMMING) return o.state = NOT_FOUND;
        const TrimKey key(o.second, o.strand, o.indexBegin, o.width, o.distance, maxED);
        auto it = P.trims.find(key);
        if (it == P.trims.end()) {
            cmb_occ oc;
            std::memset(&oc, 0, sizeof(oc));
            oc.begin = o.indexBegin, oc.end = o.indexBegin + o.width, oc.distance = o.distance, oc.strand = o.strand;
            cmb_aln al;
            std::memset(&al, 0, sizeof(al));
            al.seq_id = o.seqId, al.seq_begin = o.seqBegin, al.spans = 1, al.cigar_len = (uint16_t)o.ops.size();
            uint16_t ops[2 * 13 + 8];
            uint32_t nOps = 0;
            int found = 0;
            const std::string& pat = seqOf(o.second, o.strand);
            if (B.trimFn) {
                found = B.trimFn(B.trimUser, pairIndex, o.second, o.strand, maxED, &oc, &al, ops, 2 * 13 + 8, &nOps);
                if (found < 0) throw std::runtime_error("the trimming hook failed");
            } else {
                if (!B.textIndex) throw std::runtime_error("an occurrence runs over the end of its sequence and there is neither an index nor a hook to trim it with");
                if (cmb_trim_occurrence(B.textIndex, pat.data(), (uint32_t)pat.size(), maxED, B.metric, &oc, &al, ops, 2 * 13 + 8, &nOps, &found) != CMB_OK)
                    throw std::runtime_error(cmb_last_error());
            }
            BOcc t = o;
            if (found) {
                t.indexBegin = oc.begin, t.width = oc.end - oc.begin, t.distance = oc.distance;
                t.seqId = al.seq_id, t.seqBegin = al.seq_begin;
                t.ops.assign(ops, ops + nOps);
            }
            it = P.trims.emplace(key, std::make_pair(found != 0, t)).first;
        }
        if (!it->second.first) return o.state = NOT_FOUND;
        const BOcc& t = it->second.second;
        o.indexBegin = t.indexBegin, o.width = t.width, o.distance = t.distance, o.seqId = t.seqId, o.seqBegin = t.seqBegin, o.ops = t.ops;
        return o.state = FOUND_WITH_TRIMMING;
    }
    static BPair makePair(const BOcc& up, const BOcc& down) { // PairedTextOccs(up, down): fragment size from the assigned ranges
        BPair p;
        p.up = up, p.down = down;
        p.fragSize = (down.rangeBegin() + down.width) - up.rangeBegin();
        p.distance = up.distance + down.distance;
        return p;
    }
    // pairOccurrencesForBestMapping (:1743-1815)
    void pairOccurrencesForBestMapping(std::vector<BOcc>& U, std::vector<BOcc>& D, std::vector<BPair>& out, uint32_t uMax, uint32_t dMax,
                                       std::set<uint32_t>& uTrimmed, std::set<uint32_t>& dTrimmed) {
        if (U.empty() || D.empty()) return;
        std::stable_sort(D.begin(), D.end(), [](const BOcc& a, const BOcc& b) { return a.indexBegin < b.indexBegin; });
        for (uint32_t i = 0; i < U.size(); i++) {
            BOcc& u = U[i];
            const uint32_t upos = u.indexBegin;
            auto it = std::lower_bound(D.begin(), D.end(), upos, [](const BOcc& d, uint32_t p) { return d.indexBegin < p; });
            for (; it != D.end(); ++it) {
                const uint32_t frag = it->indexEnd() - upos;
                if (frag <= B.prm.max_frag && frag >= B.prm.min_frag) {
                    const int uf = assign(u, uMax);
                    if (uf != FOUND) {
                        if (uf == FOUND_WITH_TRIMMING) uTrimmed.insert(i);
                        break;
                    }
                    const int df = assign(*it, dMax);
                    if (df != FOUND) {
                        if (df == FOUND_WITH_TRIMMING) dTrimmed.insert((uint32_t)(it - D.begin()));
                        continue;
                    }
                    if (u.seqId != it->seqId) continue;
                    out.push_back(makePair(u, *it));
                } else if (frag > B.prm.max_frag)
                    break;
            }
        }
    }
    // handleTrimmedOccs (:814-832): an occurrence found with trimming moves to the stratum of its new distance
    static void handleTrimmedOccs(const std::set<uint32_t>& ids, uint32_t oDist, OccVector& v) {
        for (auto idIt = ids.rbegin(); idIt != ids.rend(); ++idIt) {
            BOcc occ = std::move(v[oDist].v[*idIt]);
            v[oDist].v.erase(v[oDist].v.begin() + *idIt);
            occ.state = FOUND; // removeTrimmingLabel
            if (occ.distance >= v.size()) continue; // (cannot happen: the trimmed window is verified with the stratum's bound)
            std::vector<BOcc>& t = v[occ.distance].v;
            t.insert(std::lower_bound(t.begin(), t.end(), occ, occLess), std::move(occ));
        }
    }
    static uint32_t firstPosDist(const OccVector& v) { // findFirstPosDist: the first stratum that holds something or has not been looked at
        uint32_t d = 0;
        while (d < v.size() && v[d].v.empty() && v[d].done) d++;
        return d;
    }
    // processComb (:834-912); all arithmetic in the reference's unsigned length_t
    void processComb(uint32_t uM, uint32_t uS, uint32_t dM, uint32_t dS, std::vector<BPair>& out, uint32_t totDist) {
        OccVector &U = ov[uM][uS], &D = ov[dM][dS];
        const uint32_t minDDist = firstPosDist(D), minUDist = firstPosDist(U);
        uint32_t maxUp = std::min<uint32_t>(totDist - minDDist, (uint32_t)U.size() - 1);
        uint32_t maxDown = std::min<uint32_t>(totDist - minUDist, (uint32_t)D.size() - 1);
        auto processRead = [&](uint32_t m, uint32_t s, OccVector& v, const uint32_t& max, uint32_t& maxOther) {
            if (!processSeq(m, s, max)) return false;
            const uint32_t minD = firstPosDist(v);
            maxOther = std::min<uint32_t>(totDist - minD, maxOther);
            return true;
        };
        if (maxUp <= maxDown) {
            if (!(processRead(uM, uS, U, maxUp, maxDown) && processRead(dM, dS, D, maxDown, maxUp))) return;
        } else {
            if (!(processRead(dM, dS, D, maxDown, maxUp) && processRead(uM, uS, U, maxUp, maxDown))) return;
        }
        for (uint32_t dist = minUDist + minDDist; dist <= totDist; dist++) {
            for (uint32_t uDist = minUDist; uDist <= std::min(maxUp, dist); uDist++) {
                const uint32_t dDist = dist - uDist;
                if (dDist > maxDown || dDist < minDDist) continue;
                std::set<uint32_t> uTrimmed, dTrimmed;
                pairOccurrencesForBestMapping(U[uDist].v, D[dDist].v, out, maxUp, maxDown, uTrimmed, dTrimmed);
                handleTrimmedOccs(uTrimmed, uDist, U);
                handleTrimmedOccs(dTrimmed, dDist, D);
            }
            if (!out.empty()) return;
        }
    }
    static void mergeOrMovePairs(std::vector<BPair>& p12, std::vector<BPair>& p21, std::vector<BPair>& out) { // :914-934
        if (p12.empty() || p21.empty()) {
            out = std::move(p12.empty() ? p21 : p12);
            return;
        }
        const uint32_t d12 = p12.front().distance, d21 = p21.front().distance;
        if (d12 <= d21) {
            out = std::move(p12);
            if (d12 == d21) out.insert(out.end(), p21.begin(), p21.end());
            return;
        }
        out = std::move(p21);
    }
    static bool anyBelow(const OccVector& v, uint32_t n) {
        for (uint32_t i = 0; i < std::min<uint32_t>(n, (uint32_t)v.size()); i++)
            if (!v[i].v.empty()) return true;
        return false;
    }
    // processCombFF / RF / FR (:936-1062): the two combinations of an orientation, the one that already holds matches first; the
    // second one only needs to reach what the first one found
    void processOri(std::vector<BPair>& out, uint32_t totDist, uint32_t minTotDist) {
        std::vector<BPair> pA, pB;
        struct Comb {
            uint32_t uM, uS, dM, dS;
        };
        Comb A, Bc;
        bool aFirst;
        if (B.prm.orientation == CMB_ORIENTATION_FF) {
            A = Comb{0, 0, 1, 0};  // read 1 forward upstream of read 2 forward
            Bc = Comb{1, 1, 0, 1}; // reverse complement of read 2 upstream of the reverse complement of read 1
            aFirst = anyBelow(ov[0][0], 0xFFFFFFFFu) || anyBelow(ov[1][0], 0xFFFFFFFFu);
        } else if (B.prm.orientation == CMB_ORIENTATION_RF) {
            A = Comb{0, 1, 1, 0};
            Bc = Comb{1, 1, 0, 0};
            aFirst = anyBelow(ov[0][1], minTotDist) || anyBelow(ov[1][0], minTotDist);
        } else { // FR
            A = Comb{0, 0, 1, 1};
            Bc = Comb{1, 0, 0, 1};
            aFirst = anyBelow(ov[0][0], minTotDist) || anyBelow(ov[1][1], minTotDist);
        }
        if (aFirst) {
            processComb(A.uM, A.uS, A.dM, A.dS, pA, totDist);
            totDist = pA.empty() ? totDist : pA.front().distance;
            processComb(Bc.uM, Bc.uS, Bc.dM, Bc.dS, pB, totDist);
        } else {
            processComb(Bc.uM, Bc.uS, Bc.dM, Bc.dS, pB, totDist);
            totDist = pB.empty() ? totDist : pB.front().distance;
            processComb(A.uM, A.uS, A.dM, A.dS, pA, totDist);
        }
        mergeOrMovePairs(pA, pB, out);
    }

    // ---- no concordant pair: pairDiscordantlyBest (:1664-1741) and what it calls
    void mapStratum(uint32_t m, uint32_t s, uint32_t maxD) { // searchstrategy.h:1354-1361
        Stratum& st = ov[m][s][maxD];
        if (!st.done) {
            st.v = mapRead(m, s, maxD, maxD);
            st.done = true;
        }
    }
    void addDiscPairs(std::vector<BOcc>& fw1, std::vector<BOcc>& rc1, std::vector<BOcc>& fw2, std::vector<BOcc>& rc2, uint32_t maxED) { // :1518-1585
        if ((fw1.empty() && rc1.empty()) || (fw2.empty() && rc2.empty())) return;
        auto pairOccs = [&](BOcc& a, BOcc& b) {
            if (assign(a, maxED) == NOT_FOUND || assign(b, maxED) == NOT_FOUND) return;
            const bool sameRef = a.seqId == b.seqId, aUp = a.rangeBegin() < b.rangeBegin();
            BPair p;
            p.up = aUp ? a : b, p.down = aUp ? b : a;
            p.fragSize = sameRef ? (aUp ? b.rangeBegin() + b.width - a.rangeBegin() : a.rangeBegin() + a.width - b.rangeBegin()) : 0;
            p.distance = a.distance + b.distance;
            p.discordant = true;
            pairs.push_back(std::move(p));
        };
        for (BOcc& a : fw1) {
            for (BOcc& b : fw2) pairOccs(a, b);
            for (BOcc& b : rc2) pairOccs(a, b);
        }
        for (BOcc& a : rc1) {
            for (BOcc& b : fw2) pairOccs(a, b);
            for (BOcc& b : rc2) pairOccs(a, b);
        }
    }
    // checkAlignments (:536-567)
    void checkAlignments(uint32_t m, uint32_t s, uint32_t& best, uint32_t l, uint32_t cutOff) {
        OccVector& v = ov[m][s];
        std::vector<BOcc> trimmed, assignedOccs;
        for (BOcc& o : v[l].v) {
            const int f = assign(o, cutOff);
            if (f != FOUND) {
                if (f == FOUND_WITH_TRIMMING && o.distance > l) trimmed.push_back(std::move(o));
            } else {
                assignedOccs.push_back(std::move(o));
                if (l < best) best = l;
            }
        }
        v[l].v = std::move(assignedOccs);
        for (BOcc& o : trimmed) {
            o.state = FOUND;
            if (o.distance < v.size()) v[o.distance].v.push_back(std::move(o));
        }
    }
    // findBestAlignments (:623-712) on the strata the pairing has filled so far
    bool findBestAlignments(uint32_t m, uint32_t x, uint32_t& best) {
        OccVector &fw = ov[m][0], &rc = ov[m][1];
        const uint32_t cutOff = (uint32_t)fw.size() - 1;
        best = cutOff + 1;
        bool bestFound = false;
        if (x == 0) {
            for (uint32_t s = 0; s < 2; s++)
                if (!ov[m][s][0].done) {
                    ov[m][s][0].v = mapRead(m, s, 0, 0);
                    ov[m][s][0].done = true;
                }
            if (!fw[0].v.empty() || !rc[0].v.empty()) {
                checkAlignments(m, 0, best, 0, cutOff);
                checkAlignments(m, 1, best, 0, cutOff);
                if (best == 0) bestFound = true;
            }
        }
        const uint32_t maxED = best == 0 ? x : cutOff;
        uint32_t prevK = 0;
        auto hasUpdate = [&](uint32_t s, uint32_t k) {
            if (ov[m][s][k].done) return !ov[m][s][k].v.empty();
            return processSeq(m, s, k);
        };
        for (uint32_t k = std::max(x, 1u); k <= maxED;) {
            bool update = false;
            update |= hasUpdate(0, k);
            update |= hasUpdate(1, k);
            if (update)
                for (uint32_t l = prevK + 1; l <= std::min(k, best + x); l++) {
                    checkAlignments(m, 0, best, l, maxED);
                    checkAlignments(m, 1, best, l, maxED);
                }
            if (bestFound) break;
            if (update && best < cutOff + 1) {
                bestFound = true;
                if (x == 0) break;
                prevK = k, k = std::min(best + x, maxED);
            } else {
                if (k == maxED) break;
                const uint32_t step = k < 5 ? 2 : 4;
                prevK = k;
                k = std::min(k + x + step, maxED);
            }
        }
        return bestFound;
    }
    // combineOccVectors (:569-621)
    std::vector<BOcc> combineOccVectors(uint32_t m, uint32_t best, uint32_t max) {
        auto compare = [](const BOcc& a, const BOcc& b) { return a.seqId < b.seqId || (a.seqId == b.seqId && a.rangeBegin() < b.rangeBegin()); };
        auto equal = [](const BOcc& a, const BOcc& b) { return a.seqId == b.seqId && a.rangeBegin() == b.rangeBegin(); };
        std::vector<BOcc> matches;
        for (uint32_t i = best; i <= max; i++)
            for (uint32_t s = 0; s < 2; s++) {
                std::vector<BOcc>& v = ov[m][s][i].v;
                std::stable_sort(v.begin(), v.end(), compare);
                v.erase(std::unique(v.begin(), v.end(), equal), v.end());
                matches.insert(matches.end(), v.begin(), v.end());
            }
        return matches;
    }
    std::vector<BOcc> findBestMapping(uint32_t m, uint32_t x) { // :1648-1662
        uint32_t best = 0;
        if (findBestAlignments(m, x, best)) return combineOccVectors(m, best, std::min<uint32_t>(best + x, (uint32_t)ov[m][0].size() - 1));
        return {};
    }
    void addBothUnmapped() { // searchstrategy.h:1236-1247
        if (!B.prm.unmapped_records) return;
        BPair p;
        p.upValid = p.downValid = false;
        p.up.second = 0, p.down.second = 1;
        pairs.push_back(std::move(p));
    }
    void addOneUnmapped(std::vector<BOcc>& m1, std::vector<BOcc>& m2, uint32_t maxED) { // :1463-1516
        const bool firstMapped = !m1.empty();
        for (BOcc& o : firstMapped ? m1 : m2) {
            if (assign(o, maxED) == NOT_FOUND) continue;
            BPair p;
            p.up = o;
            p.downValid = false;
            p.down.second = firstMapped ? 1 : 0;
            p.distance = o.distance;
            pairs.push_back(std::move(p));
        }
        if (pairs.empty()) addBothUnmapped();
    }
    std::vector<Unpaired> unpairedOccs;
    void addUnpairedMatches(std::vector<BOcc>& all, uint32_t m, uint32_t maxED) { // :1401-1461
        std::vector<BOcc> temp;
        for (BOcc& o : all)
            if (assign(o, maxED) != NOT_FOUND) temp.push_back(std::move(o));
        if (temp.empty()) {
            if (B.prm.unmapped_records) unpairedOccs.push_back(Unpaired{BOcc(), 0, 0, false, true, m});
            return;
        }
        std::stable_sort(temp.begin(), temp.end(), [](const BOcc& a, const BOcc& b) { return a.distance < b.distance; });
        const uint32_t best = temp.front().distance;
        const uint32_t bestCount = (uint32_t)std::count_if(temp.begin(), temp.end(), [best](const BOcc& o) { return o.distance == best; });
        bool first = true;
        for (BOcc& o : temp) {
            unpairedOccs.push_back(Unpaired{o, bestCount, best, first, false, m});
            first = false;
        }
    }
    void pairDiscordantlyBest(uint32_t x) {
        const uint32_t max1 = (uint32_t)ov[0][0].size() - 1, max2 = (uint32_t)ov[1][0].size() - 1;
        if (B.prm.discordant_allowed) {
            const uint32_t total = (uint32_t)(ov[0][0].size() + ov[1][0].size());
            uint32_t bestStratum = total + 1;
            bool bestFound = false;
            for (uint32_t i = 0; i < total; i++) {
                if (i <= max1) mapStratum(0, 0, i), mapStratum(0, 1, i);
                if (i <= max2) mapStratum(1, 0, i), mapStratum(1, 1, i);
                const uint32_t min1 = i > max2 ? i - max2 : 0;
                for (uint32_t e1 = min1; e1 <= std::min(i, max1); e1++) {
                    const uint32_t e2 = i - e1;
                    addDiscPairs(ov[0][0][e1].v, ov[0][1][e1].v, ov[1][0][e2].v, ov[1][1][e2].v, i);
                }
                if (!pairs.empty()) {
                    if (!bestFound) bestStratum = i, bestFound = true;
                    if (i == bestStratum + x) return;
                }
            }
        }
        std::vector<BOcc> best1 = findBestMapping(0, x), best2 = findBestMapping(1, x);
        if (best1.empty() && best2.empty()) addBothUnmapped();
        else if (best1.empty()) addOneUnmapped(best1, best2, max2);
        else if (best2.empty()) addOneUnmapped(best1, best2, max1);
        else {
            addUnpairedMatches(best1, 0, max1);
            addUnpairedMatches(best2, 1, max2);
        }
    }
    // matchApproxPairedEndBestPlusX (:1091-1179), without single-end results to start from
    void run() {
        const uint32_t x = B.x, cutOff1 = P.cutOff[0], cutOff2 = P.cutOff[1];
        uint32_t best = cutOff1 + cutOff2 + 1, minDistNotExplored = 0;
        if (x == 0) {
            processOri(pairs, 0, 0);
            minDistNotExplored = 1;
        }
        bool bestFound = false;
        if (!pairs.empty()) best = 0, bestFound = true;
        uint32_t maxStratum = best == 0 ? x : cutOff1 + cutOff2;
        for (uint32_t k = std::max(x, 1u); k <= maxStratum;) {
            processOri(pairs, k, minDistNotExplored);
            if (!bestFound) {
                if (!pairs.empty()) {
                    best = k;
                    bestFound = true;
                    maxStratum = std::min(best + x, cutOff1 + cutOff2);
                    minDistNotExplored = k + 1;
                    if (x == 0) break;
                    k = maxStratum;
                } else {
                    if (k == maxStratum) break;
                    const uint32_t step = k < 6 ? 2 : 4;
                    k = std::min(maxStratum, k + x + step);
                }
            } else
                break;
        }
        if (pairs.empty()) pairDiscordantlyBest(x);
    }
};

std::string callText(const std::function<int64_t(char*, uint64_t)>& f) {
    std::string s((size_t)f(nullptr, 0), '\0');
    std::vector<char> buf(s.size() + 1);
    f(buf.data(), buf.size());
    return std::string(buf.data(), s.size());
}
} // namespace

extern "C" int cmb_pair_best_create(const cmb_pair_params* prm, uint32_t x, uint32_t min_identity, uint32_t max_supported, int metric,
                                    cmb_index* text_index, uint32_t n_pairs, const cmb_pair_read* reads1, const cmb_pair_read* reads2,
                                    cmb_pair_best** out) {
    if (!prm || !out || (n_pairs && (!reads1 || !reads2)) || prm->orientation > 2) return failWith(CMB_ERR_INVALID, "bad argument");
    if (min_identity < 50 || min_identity > 100) return failWith(CMB_ERR_INVALID, "the minimal identity lies between 50 and 100");
    if (max_supported > 13) return failWith(CMB_ERR_UNSUPPORTED, "more than 13 errors (MAX_K of the reference)");
    std::unique_ptr<cmb_pair_best> b(new cmb_pair_best());
    b->prm = *prm, b->x = x, b->minIdentity = min_identity, b->maxSupported = max_supported, b->metric = metric, b->textIndex = text_index;
    b->pairs.resize(n_pairs);
    for (uint32_t i = 0; i < n_pairs; i++) {
        const cmb_pair_read* R[2] = {reads1 + i, reads2 + i};
        for (int m = 0; m < 2; m++) {
            if (!R[m]->id || !R[m]->seq || !R[m]->revcomp) return failWith(CMB_ERR_INVALID, "a read without identifier, sequence or reverse complement");
            PairState& P = b->pairs[i];
            P.id[m] = R[m]->id, P.seq[m] = R[m]->seq, P.rc[m] = R[m]->revcomp;
            P.qual[m] = R[m]->qual ? R[m]->qual : "", P.rqual[m] = R[m]->revqual ? R[m]->revqual : "";
            // getMaxED (searchstrategy.h:1797): the identity cut-off, held to what the strategy and the device support
            P.cutOff[m] = std::min<uint32_t>(max_supported, (uint32_t)((P.seq[m].size() * (100 - min_identity)) / 100));
        }
    }
    *out = b.release();
    return CMB_OK;
}

extern "C" int cmb_pair_best_set_trim(cmb_pair_best* b, cmb_pair_trim_fn fn, void* user) {
    if (!b) return failWith(CMB_ERR_INVALID, "null argument");
    b->trimFn = fn, b->trimUser = user;
    return CMB_OK;
}

// addSingleEndedForBest (searchstrategy.cpp:1064-1089) for pairSingleEndedMatchesBest (searchstrategy.h:1454-1462, x = 0): the pair starts from the
// mates' single-end BEST results (occurrences with their sequence assigned and their CIGAR)
extern "C" int cmb_pair_best_seed(cmb_pair_best* b, uint32_t pair, const cmb_occ* occ1, const cmb_aln* aln1, uint64_t n1, const uint16_t* ops1,
                                  const cmb_occ* occ2, const cmb_aln* aln2, uint64_t n2, const uint16_t* ops2, int read2_done) {
    if (!b || pair >= b->pairs.size() || (n1 && (!occ1 || !aln1)) || (n2 && (!occ2 || !aln2))) return failWith(CMB_ERR_INVALID, "bad argument");
    if (b->x != 0) return failWith(CMB_ERR_INVALID, "single-end results start a pair in BEST mode without further strata only (x = 0)");
    PairState& P = b->pairs[pair];
    if (P.finished || !P.lists[0][0].empty() || !P.lists[0][1].empty() || !P.lists[1][0].empty() || !P.lists[1][1].empty())
        return failWith(CMB_ERR_INVALID, "the pair has started its walk already");
    const cmb_occ* oc[2] = {occ1, occ2};
    const cmb_aln* al[2] = {aln1, aln2};
    const uint16_t* op[2] = {ops1, ops2};
    const uint64_t cnt[2] = {n1, n2};
    std::vector<BOcc> seeds[2];
    for (int m = 0; m < 2; m++)
        for (uint64_t j = 0; j < cnt[m]; j++) {
            const cmb_occ& o = oc[m][j];
            if (o.end < o.begin || o.strand > 1 || o.distance > P.cutOff[m]) return failWith(CMB_ERR_INVALID, "bad occurrence (or one beyond the read's cut-off)");
            if (al[m][j].spans == 1) return failWith(CMB_ERR_INVALID, "single-end results come with their sequence assigned (trimmed where they ran over its end)");
            if (al[m][j].cigar_len && !op[m]) return failWith(CMB_ERR_INVALID, "alignments without their operations");
            BOcc s;
            s.indexBegin = o.begin, s.width = o.end - o.begin, s.distance = o.distance, s.strand = (uint8_t)o.strand, s.second = (uint8_t)m;
            s.seqId = al[m][j].seq_id, s.seqBegin = al[m][j].seq_begin, s.state = FOUND;
            s.ops.assign(op[m] + al[m][j].cigar_off, op[m] + al[m][j].cigar_off + al[m][j].cigar_len);
            seeds[m].push_back(std::move(s));
        }
    P.seeds[0] = std::move(seeds[0]), P.seeds[1] = std::move(seeds[1]);
    P.seeded = true, P.read2done = read2_done != 0;
    return CMB_OK;
}

extern "C" int cmb_pair_best_cutoff(const cmb_pair_best* b, uint32_t pair, uint32_t mate, uint32_t* cut_off) {
    if (!b || !cut_off || pair >= b->pairs.size() || mate > 1) return failWith(CMB_ERR_INVALID, "bad argument");
    *cut_off = b->pairs[pair].cutOff[mate];
    return CMB_OK;
}

extern "C" int cmb_pair_best_advance(cmb_pair_best* b, cmb_pair_request* req, uint64_t cap, uint64_t* n) {
    if (!b || !n || (cap && !req)) return failWith(CMB_ERR_INVALID, "null argument");
    *n = 0;
    try {
        for (uint32_t i = 0; i < b->pairs.size(); i++) {
            PairState& P = b->pairs[i];
            if (P.finished) continue;
            Walk w(*b, i);
            try {
                w.run();
                P.finished = true;
                P.pairs = std::move(w.pairs);
                P.unpaired = std::move(w.unpairedOccs);
            } catch (const Need& need) {
                if (*n < cap) req[*n] = cmb_pair_request{i, need.mate, need.strand, need.k};
                (*n)++;
            }
        }
    } catch (const std::exception& e) {
        return failWith(CMB_ERR_DEVICE, e.what());
    }
    if (*n > cap) return failWith(CMB_ERR_OVERFLOW, "room for the requests (one per unfinished pair at most)");
    return CMB_OK;
}

extern "C" int cmb_pair_best_supply(cmb_pair_best* b, uint32_t pair, uint32_t mate, uint32_t strand, uint32_t max_distance, const cmb_occ* occ,
                                    const cmb_aln* aln, uint64_t n_occ, const uint16_t* cigar_ops) {
    if (!b || pair >= b->pairs.size() || mate > 1 || strand > 1 || (n_occ && (!occ || !aln))) return failWith(CMB_ERR_INVALID, "bad argument");
    PairState& P = b->pairs[pair];
    if (max_distance > P.cutOff[mate]) return failWith(CMB_ERR_INVALID, "a list beyond the read's cut-off");
    std::vector<BOcc> v;
    for (uint64_t j = 0; j < n_occ; j++) {
        if (occ[j].strand != strand) continue; // (a list may hold both strands of the read: the other one is supplied by its own call)
        if (occ[j].end < occ[j].begin || occ[j].distance > max_distance) return failWith(CMB_ERR_INVALID, "bad occurrence");
        BOcc o;
        o.indexBegin = occ[j].begin, o.width = occ[j].end - occ[j].begin, o.distance = occ[j].distance;
        o.strand = (uint8_t)strand, o.second = (uint8_t)mate;
        o.seqId = aln[j].seq_id, o.seqBegin = aln[j].seq_begin, o.spans = aln[j].spans ? 1 : 0;
        if (aln[j].cigar_len && !cigar_ops) return failWith(CMB_ERR_INVALID, "alignments without their operations");
        o.ops.assign(cigar_ops + aln[j].cigar_off, cigar_ops + aln[j].cigar_off + aln[j].cigar_len);
        v.push_back(std::move(o));
    }
    if (max_distance == 0) std::stable_sort(v.begin(), v.end(), occLess); // mapRead sorts the exact matches (searchstrategy.h:499-501)
    P.lists[mate][strand][max_distance] = std::move(v);
    return CMB_OK;
}

extern "C" int64_t cmb_pair_best_sam(const cmb_pair_best* b, uint32_t pair, const char* const* seq_names, char* out, uint64_t cap, uint32_t* n_pairs_out) {
    if (!b || !seq_names || pair >= b->pairs.size()) return failWith(CMB_ERR_INVALID, "bad argument");
    const PairState& P = b->pairs[pair];
    if (!P.finished) return failWith(CMB_ERR_INVALID, "the pair still waits for a list (cmb_pair_best_advance)");
    std::vector<BPair> pairs = P.pairs;
    auto hitFrom = [&](const BOcc& o) {
        cmb_sam_hit h;
        h.seq_name = seq_names[o.seqId];
        h.pos0 = o.seqBegin, h.distance = o.distance, h.revcomp = o.strand;
        h.cigar_ops = o.ops.data(), h.n_ops = (uint32_t)o.ops.size();
        return h;
    };
    auto printSeq = [&](const BOcc& o) { return (o.strand ? P.rc[o.second] : P.seq[o.second]).c_str(); };
    auto printQual = [&](const BOcc& o) { return (o.strand ? P.rqual[o.second] : P.qual[o.second]).c_str(); };
    auto unmappedLine = [&](int m, bool mateMapped, bool mateRev) {
        return callText([&](char* o, uint64_t c) { return cmb_sam_unmapped_pe(P.id[m].c_str(), P.seq[m].c_str(), P.qual[m].c_str(), m == 0, mateMapped, mateRev, o, c); });
    };
    // generateSAMPairedEnd (:1904-1970): the first pair of minimal distance becomes the primary one
    uint32_t nPairs = 0;
    if (!pairs.empty()) {
        size_t mi = 0;
        for (size_t i = 1; i < pairs.size(); i++)
            if (pairs[i].distance < pairs[mi].distance) mi = i;
        const uint32_t bestScore = pairs[mi].distance;
        for (const BPair& p : pairs) nPairs += p.distance == bestScore;
        if (mi != 0) std::swap(pairs[0], pairs[mi]);
        bool primary = true;
        for (BPair& p : pairs) {
            for (int side = 0; side < 2; side++) {
                const BOcc& me = side ? p.down : p.up;
                const BOcc& mate = side ? p.up : p.down;
                const bool meValid = side ? p.downValid : p.upValid, mateValid = side ? p.upValid : p.downValid;
                std::string& line = side ? p.downLine : p.upLine;
                if (!meValid) { // createUnmappedSAMOccurrencePE: written when the pair was made
                    line = unmappedLine(me.second, mateValid, mateValid && mate.strand != 0);
                    continue;
                }
                const cmb_sam_hit h = hitFrom(me);
                cmb_sam_hit mh;
                if (mateValid) mh = hitFrom(mate);
                line = callText([&](char* o, uint64_t c) {
                    return cmb_sam_pe(P.id[me.second].c_str(), &h, !me.second, mateValid ? &mh : nullptr, nPairs, bestScore, p.fragSize, p.discordant, primary,
                                      printSeq(me), printQual(me), o, c);
                });
            }
            primary = false;
        }
    }
    // OutputWriter::writeChunks (fastq.cpp:662-702)
    std::string text;
    const bool mapped = !pairs.empty() && pairs.front().upValid && pairs.front().downValid;
    const bool mappedHalf = !mapped && !pairs.empty() && (pairs.front().upValid || pairs.front().downValid);
    bool firstWrite = true;
    for (const BPair& p : pairs) {
        text += p.upLine;
        if (!mappedHalf || firstWrite) text += p.downLine;
        firstWrite = false;
    }
    for (const Unpaired& u : P.unpaired) {
        if (u.unmapped) {
            text += unmappedLine((int)u.mate, false, false);
            continue;
        }
        const cmb_sam_hit h = hitFrom(u.o);
        text += callText([&](char* o, uint64_t c) {
            return cmb_sam_unpaired(P.id[u.mate].c_str(), &h, u.mate == 0, u.bestCount, u.best, u.first, printSeq(u.o), printQual(u.o), o, c);
        });
    }
    if (n_pairs_out) *n_pairs_out = mapped ? (uint32_t)pairs.size() : 0; // TOTAL_UNIQUE_PAIRS
    if (out && cap > text.size()) std::memcpy(out, text.c_str(), text.size() + 1);
    return (int64_t)text.size();
}

extern "C" void cmb_pair_best_destroy(cmb_pair_best* b) {
    delete b;
}

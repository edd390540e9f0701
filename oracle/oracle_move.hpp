// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_core.hpp for the rules).
//
// CPU restatement of the run-length compressed backend of biointec/columba
// v2.0.3 (b-move; RUN_LENGTH_COMPRESSION flavour without PHI_MOVE, ALPHABET = 5;
// templated on length_t: 64 bits is that flavour's default, CMakeLists.txt:41-63): the bit-packed move table, the run walks, LF with
// fast-forward, character extension with toehold maintenance, and locate by
// phi / phi^-1 bounded by the PLCP array.
//
// Pinning status:
//   * MoveLF (bmove/moverepr.{h,cpp}: row packing, file layout, getRunIndex /
//     computeRunIndices, walkToNextRun / walkToPreviousRun, fastForward, findLF,
//     addChar, countChar, getCumulativeCounts) and MoveRange
//     (indexhelpers.h:137-255): PINNED against the reference's own moverepr.cpp
//     compiled unmodified with -DRUN_LENGTH_COMPRESSION into
//     oracle/_ref/ref_driver_rlc64 and _rlc32 (tests/golden/ref_vectors_rlc.*).
//   * BMove (bmove/bmove.{h,cpp}: the three extension variants, toeholds, phi,
//     collectTextPositions) and PLCP (bmove/plcp.h): the reference units need
//     sdsl-lite, which is absent from this image: PARITY UNPINNED, checked by
//     brute force against the suffix array (tests/test_move_oracle.py).
//
// Paths relative to /root/reference/src.
// ============================================================================
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace orc {

// indexhelpers.h:137-255 (MoveRange = SARange of the RLC flavour)
template <typename L> struct MoveRangeT {
    L begin = 0, end = 0, beginRun = 0, endRun = 0;
    bool runIndicesValid = true;
    MoveRangeT() {}
    MoveRangeT(L b, L e, L br, L er, bool v = true) : begin(b), end(e), beginRun(br), endRun(er), runIndicesValid(v) {}
    bool empty() const { return end <= begin; }            // indexhelpers.h:94-96
    L width() const { return empty() ? 0 : end - begin; } // indexhelpers.h:102-104
    void setEmpty() { begin = end = beginRun = endRun = 0; runIndicesValid = false; } // :222-228
};

// bmove/moverepr.h:86-241 (MoveLFReprBP on top of BitPackedRepresentation)
template <typename L> class MoveLFT {
    typedef MoveRangeT<L> MoveRange;
  // bmove/moverepr.h:35-80, moverepr.cpp:36-64 (BitPackedRepresentation)
    std::vector<uint8_t> buffer;
    L nrOfRuns = 0, textSize = 0;
    uint8_t bitsForN = 0, bitsForR = 0;
    uint16_t totalBits = 0, totalBytes = 0;

    static uint8_t bitsFor(double v) { return (uint8_t)std::ceil(std::log2(v)); } // moverepr.h:44-46

  public:
    L size() const { return nrOfRuns; }
    L getTextSize() const { return textSize; }
    uint16_t rowBytes() const { return totalBytes; }
    uint8_t nBits() const { return bitsForN; }
    uint8_t rBits() const { return bitsForR; }

    L getRowValue(L rowIndex, uint16_t bitOffset, uint8_t numBits) const { // moverepr.cpp:36-47
        L mask = (L)((1ULL << numBits) - 1);
        uint64_t byteIndex = (uint64_t)rowIndex * totalBytes + bitOffset / 8;
        unsigned __int128 v;
        std::memcpy(&v, &buffer[byteIndex], 16);
        return (L)(v >> (bitOffset % 8)) & mask;
    }
    void setRowValue(L rowIndex, L value, uint16_t bitOffset, uint8_t numBits) { // moverepr.cpp:49-64
        L mask = (L)((1ULL << numBits) - 1);
        value &= mask;
        uint64_t byteIndex = (uint64_t)rowIndex * totalBytes + bitOffset / 8;
        unsigned __int128 v;
        std::memcpy(&v, &buffer[byteIndex], 16);
        v &= ~((unsigned __int128)mask << (bitOffset % 8));
        v |= (unsigned __int128)value << (bitOffset % 8);
        std::memcpy(&buffer[byteIndex], &v, 16);
    }

  private:
    uint8_t bitsForC = 0;
    L zeroCharPos = 0;

    void layout() { // moverepr.cpp:75-84, :113-121
        bitsForC = bitsFor(5.0);
        bitsForN = bitsFor((double)textSize);
        bitsForR = bitsFor((double)nrOfRuns);
        totalBits = (uint16_t)(bitsForC + 2 * bitsForN + bitsForR);
        totalBytes = (uint16_t)((totalBits + 7) / 8);
        buffer.assign((uint64_t)totalBytes * ((uint64_t)nrOfRuns + 1) + 16, 0);
    }

  public:
    bool initialize(L runs, L n) { // moverepr.cpp:70-101
        nrOfRuns = runs;
        textSize = n;
        layout();
        return true;
    }
    // moverepr.cpp:103-143: the bytes of a .LFBP file
    bool loadBytes(const uint8_t* p, size_t len) {
        const size_t W = sizeof(L);
        if (len < 3 * W) return false;
        std::memcpy(&textSize, p, W);
        std::memcpy(&nrOfRuns, p + W, W);
        std::memcpy(&zeroCharPos, p + 2 * W, W);
        layout();
        uint64_t body = (uint64_t)totalBytes * ((uint64_t)nrOfRuns + 1);
        if (len < 3 * W + body) return false;
        std::memcpy(buffer.data(), p + 3 * W, body);
        return true;
    }
    std::vector<uint8_t> fileBytes() const { // moverepr.cpp:145-168
        uint64_t body = (uint64_t)totalBytes * ((uint64_t)nrOfRuns + 1);
        const size_t W = sizeof(L);
        std::vector<uint8_t> out(3 * W + body);
        std::memcpy(&out[0], &textSize, W);
        std::memcpy(&out[W], &nrOfRuns, W);
        std::memcpy(&out[2 * W], &zeroCharPos, W);
        std::memcpy(&out[3 * W], buffer.data(), body);
        return out;
    }
    void setRowValues(L row, uint8_t runChar, L inStart, L outStart, L outRun) { // moverepr.cpp:170-181
        setRowValue(row, runChar, 0, bitsForC);
        setRowValue(row, inStart, bitsForC, bitsForN);
        setRowValue(row, outStart, bitsForC + bitsForN, bitsForN);
        setRowValue(row, outRun, bitsForC + 2 * bitsForN, bitsForR);
    }
    uint8_t getRunHead(L i) const { return (uint8_t)getRowValue(i, 0, bitsForC); }                     // :183-187
    L getInputStartPos(L i) const { return getRowValue(i, bitsForC, bitsForN); }                  // :189-193
    L getOutputStartPos(L i) const { return getRowValue(i, bitsForC + bitsForN, bitsForN); }      // :195-199
    L getOutputStartRun(L i) const { return getRowValue(i, bitsForC + 2 * bitsForN, bitsForR); }  // :201-205
    void setOutputStartRun(L row, L v) { setRowValue(row, v, bitsForC + 2 * bitsForN, bitsForR); } // :207-211
    void setZeroCharPos(L v) { zeroCharPos = v; }
    L getZeroCharPos() const { return zeroCharPos; }

    void getRunIndex(L position, L& runIndex, std::pair<L, L>& possible) const { // moverepr.cpp:213-234
        while (possible.second > possible.first) {
            L test = (possible.first + possible.second + 1) / 2;
            if (getInputStartPos(test) <= position) possible.first = test;
            else possible.second = test - 1;
        }
        runIndex = possible.first;
    }
    void computeRunIndices(MoveRange& r) const { // moverepr.cpp:236-249
        L begin = r.begin, end = r.end - 1, beginRun = r.beginRun, endRun = r.endRun;
        std::pair<L, L> possible(beginRun, endRun);
        getRunIndex(begin, beginRun, possible);
        r.beginRun = beginRun;
        possible.second = endRun;
        getRunIndex(end, endRun, possible);
        r.endRun = endRun;
        r.runIndicesValid = true;
    }
    bool walkToNextRun(const MoveRange& r, L& nextPos, L& nextRun, L c, uint64_t* steps = nullptr) const { // :251-266
        nextPos = r.begin;
        nextRun = r.beginRun;
        while (getRunHead(nextRun) != c && nextRun <= r.endRun) {
            nextRun++;
            nextPos = getInputStartPos(nextRun);
            if (steps) ++*steps;
        }
        return nextRun <= r.endRun;
    }
    void walkToPreviousRun(const MoveRange& r, L& prevPos, L& prevRun, L c, uint64_t* steps = nullptr) const { // :268-281
        prevPos = r.end - 1;
        prevRun = r.endRun;
        while (getRunHead(prevRun) != c) {
            prevPos = getInputStartPos(prevRun) - 1;
            prevRun--;
            if (steps) ++*steps;
        }
    }
    void fastForward(L position, L& runIndex, uint64_t* steps = nullptr) const { // :283-293
        while (getInputStartPos(runIndex) <= position) {
            runIndex++;
            if (steps) ++*steps;
        }
        runIndex--;
    }
    void findLF(L& position, L& runIndex, uint64_t* steps = nullptr) const { // :295-301
        L offset = position - getInputStartPos(runIndex);
        position = getOutputStartPos(runIndex) + offset;
        runIndex = getOutputStartRun(runIndex);
        fastForward(position, runIndex, steps);
    }
    void findLFWithoutFastForward(L& position, L runIndex) const { // :303-307
        position = getOutputStartPos(runIndex) + (position - getInputStartPos(runIndex));
    }
    void addChar(const MoveRange& parent, MoveRange& child, L c, uint64_t* steps = nullptr) const { // :309-327
        L nextPos, nextRun;
        if (!walkToNextRun(parent, nextPos, nextRun, c, steps)) {
            child.setEmpty();
            return;
        }
        L prevPos, prevRun;
        walkToPreviousRun(parent, prevPos, prevRun, c, steps);
        findLF(nextPos, nextRun, steps);
        findLF(prevPos, prevRun, steps);
        child = MoveRange(nextPos, prevPos + 1, nextRun, prevRun);
    }
    L countChar(const MoveRange& parent, L c, uint64_t* steps = nullptr) const { // :329-345
        L nextPos, nextRun;
        if (!walkToNextRun(parent, nextPos, nextRun, c, steps)) return 0;
        L prevPos, prevRun;
        walkToPreviousRun(parent, prevPos, prevRun, c, steps);
        findLFWithoutFastForward(nextPos, nextRun);
        findLFWithoutFastForward(prevPos, prevRun);
        return prevPos + 1 - nextPos;
    }
    L getCumulativeCounts(const MoveRange& r, L posInAlphabet, uint64_t* steps = nullptr) const { // :347-365
        L cum = 0;
        if (r.begin <= zeroCharPos && r.end > zeroCharPos) cum++;
        for (L c = 1; c < posInAlphabet; c++) cum += countChar(r, c, steps);
        return cum;
    }
};

// The construction loop of buildindex.cpp:826-915 (fillRows + createAndWriteMove) over any class with MoveLF's
// interface — instantiated with MoveLF here and with the reference's MoveLFReprBP in ref_driver_rlc.cpp.
// bwt: codes 0..4 ($ACGT); cumCounts[c] = number of characters smaller than c.
template <typename L, class Rows> void buildMoveRows(const std::vector<uint8_t>& bwt, const L cumCounts[5], Rows& rows) {
    const L n = (L)bwt.size();
    std::vector<L> inStart, outStart;
    std::vector<uint8_t> head;
    L seen[5] = {0, 0, 0, 0, 0}, zeroPos = n;
    int prev = 5;
    for (L i = 0; i < n; i++) {
        int c = bwt[i];
        if (c == 0) zeroPos = i;
        if (c != prev) {
            inStart.push_back(i);
            head.push_back((uint8_t)c);
            outStart.push_back(cumCounts[c] + seen[c]);
        }
        seen[c]++;
        prev = c;
    }
    const L r = (L)inStart.size();
    rows.setZeroCharPos(zeroPos);
    rows.initialize(r, n);
    for (L i = 0; i < r; i++) rows.setRowValues(i, head[i], inStart[i], outStart[i], 0);
    for (L i = 0; i < r; i++) { // getRunIndex of buildindex.cpp:791-815
        L lo = 0, hi = r - 1;
        while (hi - lo >= 1) {
            L t = (hi + lo) / 2 + 1;
            if (rows.getInputStartPos(t) <= outStart[i]) lo = t;
            else hi = t - 1;
        }
        rows.setOutputStartRun(i, lo);
    }
    rows.setRowValues(r, 0, n, n, r);
}

// buildindex.cpp:942-953 (buildSamples)
template <typename L> void buildSamples(const std::vector<L>& sa, const std::vector<uint8_t>& bwt, std::vector<L>& first,
                                        std::vector<L>& last) {
    first.clear();
    last.clear();
    first.push_back(sa[0]);
    for (size_t p = 0; p + 1 < bwt.size(); p++)
        if (bwt[p] != bwt[p + 1]) {
            last.push_back(sa[p]);
            first.push_back(sa[p + 1]);
        }
    last.push_back(sa[bwt.size() - 1]);
}

// indexhelpers.h:1040-1111 (ToeholdInterface) + :1117-1260 (SARangePair of the RLC flavour)
template <typename L> struct MovePairT {
    typedef MoveRangeT<L> MoveRange;
    MoveRange sa, rev;
    L toehold = 0;
    bool toeholdRepresentsEnd = false;
    L originalDepth = 0;
    MovePairT() {}
    MovePairT(const MoveRange& a, const MoveRange& b, L t, bool e, L d) : sa(a), rev(b), toehold(t), toeholdRepresentsEnd(e), originalDepth(d) {}
    L width() const { return sa.width(); }
    bool empty() const { return sa.empty(); }
};

struct MoveCounters {
    uint64_t rowSteps = 0; // rows stepped over by run walks and fast-forwards (the byte model's unit for this backend)
};

// bmove/bmove.h, bmove/bmove.cpp (without PHI_MOVE)
template <typename L> class BMoveIndexT {
  public:
    typedef MoveRangeT<L> MoveRange;
    typedef MovePairT<L> MovePair;
    typedef MoveLFT<L> MoveLF;
    L textLength = 0;
    MoveLF move, moveR;
    std::vector<L> samplesFirst, samplesLast, revSamplesFirst, revSamplesLast;
    // locate: the sorted marked positions of predFirst / predLast (buildindex.cpp:990-1013) with firstToRun / lastToRun
    // (buildindex.cpp:1044-1066), and the PLCP array (bmove/plcp.h; held plain here, operator[] = the array element)
    std::vector<L> predFirst, predLast, firstToRun, lastToRun, plcp;
    // rows stepped over by the walks, per thread (the matcher runs one thread per read shard on the same index)
    static uint64_t* rowStepsPtr() {
        static thread_local MoveCounters c;
        return &c.rowSteps;
    }

    L getInitialToehold() const { return samplesLast.back() - 1; } // bmove.h:139-142
    MovePair getCompleteRange() const {                                    // bmove.h:369-373
        return MovePair(MoveRange(0, textLength, 0, move.size() - 1), MoveRange(0, textLength, 0, moveR.size() - 1),
                        getInitialToehold(), false, 0);
    }

    L computeToeholdOn(const MoveLF& m, const std::vector<L>& first, const std::vector<L>& last, const MoveRange& r,
                            L c) const { // bmove.cpp:222-266 (computeToehold / computeToeholdRev)
        if (m.getRunHead(r.endRun) == c) return first[r.endRun] - 1;
        L prevPos, prevRun;
        m.walkToPreviousRun(r, prevPos, prevRun, c, rowStepsPtr());
        return last[prevRun] - 1;
    }

    // bmove.cpp:328-382 (findRangesWithExtraCharBackward)
    bool extendBackward(L c, const MovePair& parent, MovePair& child) const {
        MoveRange range1, trivial = parent.sa;
        if (!trivial.runIndicesValid) move.computeRunIndices(trivial); // bmove.cpp:289-297
        move.addChar(trivial, range1, c, rowStepsPtr());
        if (range1.empty()) {
            child = MovePair(range1, range1, 0, false, 0);
            return false;
        }
        const MoveRange& other = parent.rev;
        if (trivial.width() == range1.width()) {
            child = MovePair(range1, other, parent.toehold - !parent.toeholdRepresentsEnd, parent.toeholdRepresentsEnd,
                             parent.originalDepth + 1);
            return true;
        }
        L s = parent.rev.begin;
        L x = move.getCumulativeCounts(trivial, c, rowStepsPtr());
        MoveRange range2(s + x, s + x + range1.width(), other.beginRun, other.endRun);
        range2.runIndicesValid = false;
        L newToehold = computeToeholdOn(move, samplesFirst, samplesLast, trivial, c);
        child = MovePair(range1, range2, newToehold, false, parent.originalDepth + 1);
        return true;
    }
    // bmove.cpp:384-442 (findRangesWithExtraCharForward)
    bool extendForward(L c, const MovePair& parent, MovePair& child) const {
        MoveRange trivial = parent.rev;
        if (!trivial.runIndicesValid) moveR.computeRunIndices(trivial);
        MoveRange range1;
        moveR.addChar(trivial, range1, c, rowStepsPtr());
        if (range1.empty()) {
            child = MovePair(range1, range1, 0, false, 0);
            return false;
        }
        const MoveRange& other = parent.sa;
        if (trivial.width() == range1.width()) {
            child = MovePair(other, range1, parent.toehold + parent.toeholdRepresentsEnd, parent.toeholdRepresentsEnd,
                             parent.originalDepth + 1);
            return true;
        }
        L s = parent.sa.begin;
        L x = moveR.getCumulativeCounts(trivial, c, rowStepsPtr());
        MoveRange range2(s + x, s + x + range1.width(), other.beginRun, other.endRun);
        range2.runIndicesValid = false;
        L newToehold = textLength - 1 - computeToeholdOn(moveR, revSamplesFirst, revSamplesLast, trivial, c);
        child = MovePair(range2, range1, newToehold, true, parent.originalDepth + 1);
        return true;
    }
    // bmove.cpp:444-478 (findRangesWithExtraCharBackwardUniDirectional)
    bool extendBackwardUni(L c, const MovePair& parent, MovePair& child) const {
        MoveRange range1, trivial = parent.sa;
        if (!trivial.runIndicesValid) move.computeRunIndices(trivial);
        move.addChar(trivial, range1, c, rowStepsPtr());
        if (range1.empty()) {
            child = MovePair(range1, range1, 0, false, 0);
            return false;
        }
        if (trivial.width() == range1.width()) {
            child = MovePair(range1, MoveRange(), parent.toehold - !parent.toeholdRepresentsEnd, parent.toeholdRepresentsEnd,
                             parent.originalDepth + 1);
            return true;
        }
        L newToehold = computeToeholdOn(move, samplesFirst, samplesLast, trivial, c);
        child = MovePair(range1, MoveRange(), newToehold, false, parent.originalDepth + 1);
        return true;
    }
    // bmove.cpp:299-326 (findRangeWithExtraCharBackward: one range with its toehold)
    bool extendRangeBackward(L c, MoveRange& range, L& toehold, bool& repEnd, L& depth) const {
        MoveRange range1, trivial = range;
        if (!trivial.runIndicesValid) move.computeRunIndices(trivial);
        move.addChar(trivial, range1, c, rowStepsPtr());
        if (range1.empty()) {
            range = range1;
            toehold = 0, repEnd = false, depth = 0;
            return false;
        }
        if (trivial.width() == range1.width()) {
            toehold = toehold - !repEnd;
        } else {
            toehold = computeToeholdOn(move, samplesFirst, samplesLast, trivial, c);
            repEnd = false;
        }
        range = range1;
        depth = depth + 1;
        return true;
    }
    // indexinterface.cpp:947-1014 (exactMatchesOutput, RUN_LENGTH_COMPRESSION branch): begin positions of the exact
    // occurrences of s (codes 1..4; anything else: not in the alphabet -> no occurrence), in the order of
    // collectTextPositions; nodes = NODE_COUNTER increments
    void exactMatches(const std::vector<int>& s, std::vector<L>& positions, uint64_t& nodes) const {
        if (s.empty()) return;
        const MovePair all = getCompleteRange();
        MoveRange range = all.sa;
        L toehold = all.toehold, depth = all.originalDepth;
        bool repEnd = all.toeholdRepresentsEnd;
        for (size_t i = s.size(); i-- > 0;) {
            if (s[i] < 1 || s[i] > 4 || !extendRangeBackward((L)s[i], range, toehold, repEnd, depth)) return;
            nodes++;
        }
        MovePair p(range, MoveRange(), toehold, repEnd, depth);
        locate(p, positions); // getBeginPositions (bmove.cpp:562-575)
    }
    // indexinterface.cpp:294-335 (populateTable, RLC branch): the ranges of every k-mer after `wordSize` forward extensions
    // from the complete range, run indices of the SA range made exact (updateRangeSARuns -> computeRunIndices, bmove.cpp:272-274);
    // key = 2 bits per character, first character in the highest bits; k-mers that do not occur keep SARangePair()
    std::vector<MovePair> kmerTable(unsigned wordSize) const {
        std::vector<MovePair> table((size_t)1 << (2 * wordSize));
        for (size_t key = 0; key < table.size(); key++) {
            MovePair cur = getCompleteRange(), next;
            bool ok = true;
            for (int i = (int)wordSize - 1; i >= 0 && ok; i--) {
                ok = extendForward((L)((key >> (2 * i)) & 3) + 1, cur, next);
                cur = next;
            }
            if (!ok) continue;
            move.computeRunIndices(cur.sa);
            table[key] = cur;
        }
        return table;
    }
    bool extend(int mode, L c, const MovePair& parent, MovePair& child) const {
        return mode == 0 ? extendForward(c, parent, child) : mode == 1 ? extendBackward(c, parent, child) : extendBackwardUni(c, parent, child);
    }

    // sparsebitvec.h:100-102, :131-133 on a sorted list of marked positions: rank(i) = marks < i
    static L predRankCircular(const std::vector<L>& marks, L pos) {
        L rk = (L)(std::lower_bound(marks.begin(), marks.end(), pos) - marks.begin());
        return rk == 0 ? (L)marks.size() - 1 : rk - 1;
    }
    void phi(L& pos) const { // bmove.cpp:178-196
        L predRank = predRankCircular(predFirst, pos);
        L pred = predFirst[predRank];
        L delta = pred < pos ? pos - pred : pos + 1;
        L prevSample = samplesLast[firstToRun[predRank] - 1];
        pos = (L)(((uint64_t)prevSample + delta - 1) % textLength);
    }
    void phiInverse(L& pos) const { // bmove.cpp:198-217
        L predRank = predRankCircular(predLast, pos);
        L pred = predLast[predRank];
        L delta = pred < pos ? pos - pred : pos + 1;
        L nextSample = samplesFirst[lastToRun[predRank] + 1];
        pos = (L)(((uint64_t)nextSample + delta - 1) % textLength);
    }
    // bmove.cpp:500-541 (collectTextPositions), :543-560 (getTextPositionsFromSARange)
    void locate(const MovePair& ranges, std::vector<L>& positions) const {
        L firstPos = ranges.toehold - (ranges.toeholdRepresentsEnd ? ranges.originalDepth - 1 : 0);
        L depth = ranges.originalDepth;
        L cur = firstPos;
        positions.push_back(cur);
        while (plcp[cur] >= depth) {
            phi(cur);
            positions.push_back(cur);
        }
        cur = firstPos;
        while (cur != getInitialToehold() + 1) {
            phiInverse(cur);
            if (plcp[cur] < depth) break;
            positions.push_back(cur);
        }
    }
};

// MoveRange::operator== (indexhelpers.h:243-246), ToeholdInterface::operator== (:1098-1102), SARangePair::operator== (:1226-1233)
template <typename L> inline bool operator==(const MoveRangeT<L>& a, const MoveRangeT<L>& b) {
    return a.begin == b.begin && a.end == b.end && a.beginRun == b.beginRun && a.endRun == b.endRun;
}
template <typename L> inline bool operator==(const MovePairT<L>& a, const MovePairT<L>& b) {
    return a.sa == b.sa && a.toehold == b.toehold && a.toeholdRepresentsEnd == b.toeholdRepresentsEnd &&
           a.originalDepth == b.originalDepth;
}
template <typename L> inline L saBeginOf(const MovePairT<L>& r) { return r.sa.begin; }

typedef MoveRangeT<uint64_t> MoveRange64;
typedef MoveLFT<uint64_t> MoveLF64;
typedef MoveLFT<uint32_t> MoveLF32;
typedef MovePairT<uint64_t> MovePair64;
typedef BMoveIndexT<uint64_t> BMoveIndex64;

} // namespace orc

// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_core.hpp header).
//
// The search layer (oracle_search.hpp: MatcherT) over the run-length compressed backend: what the reference compiles
// with -DRUN_LENGTH_COMPRESSION (BMove instead of FMIndex behind the same IndexInterface, src/bmove/bmove.{h,cpp}).
// This adapter gives MatcherT the index operations of that flavour; the RUN_LENGTH_COMPRESSION branches of
// indexinterface.cpp / searchstrategy.cpp themselves are the `if constexpr (RLC)` sites of oracle_search.hpp.
// PARITY UNPINNED (bmove.cpp needs sdsl-lite, indexinterface.cpp parallel_hashmap): checked by brute force and against
// the FM-index restatement with in-text verification switched off (tests/test_move_search_oracle.py) — the two flavours
// walk the same search tree over the same suffix-array intervals.
// ============================================================================
#pragma once
#include "oracle_move.hpp"
#include "oracle_search.hpp"

namespace orc {

struct MoveIndexAdapter {
    typedef MovePair64 RangePair;
    typedef MoveRange64 SARange;
    static constexpr bool RLC = true;
    const BMoveIndex64& bm;
    len_t textLength = 0;
    len_t switchPoint = 0; // BMove::getSwitchPoint (bmove.cpp:195-197)
    len_t wordSize = 10;
    std::vector<MovePair64> kmerTable; // populateTable, RLC flavour (BMoveIndexT::kmerTable)
    std::vector<len_t> seqStarts;
    // the text beside the index (tests only: BEST mode's CIGARs and trimming read the occurrence's matched string from it)
    const uint8_t* text = nullptr;
    std::vector<uint8_t> textCopy;
    void attachText(const uint8_t* t, uint64_t n, const uint32_t* starts, uint32_t nStarts) { // starts: as Index::seqStarts
        textCopy.assign(t, t + n);
        text = textCopy.data();
        seqStarts.assign(starts, starts + nStarts);
    }

    MoveIndexAdapter(const BMoveIndex64& b, len_t ws) : bm(b), textLength((len_t)b.textLength), wordSize(ws) {
        kmerTable = b.kmerTable(ws);
    }
    static int c2i(char c) { return Index::c2i(c); }
    static char i2c(int i) { return Index::i2c(i); }

    RangePair completeRange() const { return bm.getCompleteRange(); }
    bool extendBackward(len_t c, const RangePair& p, RangePair& child) const { return bm.extendBackward(c, p, child); }
    bool extendForward(len_t c, const RangePair& p, RangePair& child) const { return bm.extendForward(c, p, child); }
    bool extendBackwardUni(len_t c, const RangePair& p, RangePair& child) const { return bm.extendBackwardUni(c, p, child); }

    // SARangeBackwards of this flavour: a range with its toehold (indexhelpers.h:1243-1262)
    struct ExactRange {
        MoveRange64 range;
        uint64_t toehold = 0, depth = 0;
        bool repEnd = false;
    };
    ExactRange exactStart() const { // indexinterface.cpp:958-961
        const MovePair64 all = bm.getCompleteRange();
        ExactRange r;
        r.range = all.sa, r.toehold = all.toehold, r.repEnd = all.toeholdRepresentsEnd, r.depth = all.originalDepth;
        return r;
    }
    bool extendExact(len_t c, ExactRange& r) const { // findRangeWithExtraCharBackward (bmove.cpp:299-326)
        return bm.extendRangeBackward(c, r.range, r.toehold, r.repEnd, r.depth);
    }
    std::vector<len_t> beginPositionsExact(const ExactRange& r, Counters& cnt) const { // getBeginPositions (bmove.cpp:562-575)
        MovePair64 p(r.range, MoveRange64(), r.toehold, r.repEnd, r.depth);
        return textPositions(p, cnt);
    }
    // getTextPositionsFromSARange (bmove.cpp:543-560)
    std::vector<len_t> textPositions(const RangePair& r, Counters& cnt) const {
        std::vector<uint64_t> pos;
        bm.locate(r, pos);
        if (pos.size() != r.sa.width()) throw std::runtime_error("oracle: phi chain and range width disagree");
        cnt.inc(LOCATED_ROWS, pos.size());
        cnt.inc(LF_STEPS, pos.size() + 1); // phi / phi^-1 steps: one per further position, plus the two that end the chains
        return std::vector<len_t>(pos.begin(), pos.end());
    }
    // BMove::getRangeOfSingleChar (bmove.cpp:484-497): the complete range extended backward by the character
    RangePair rangeOfSingleChar(char c) const {
        int i = c2i(c);
        if (i < 0) return RangePair();
        RangePair pair = bm.getCompleteRange(), child;
        bm.extendBackward((uint64_t)i, pair, child);
        return child;
    }
    // indexinterface.h:590-594
    RangePair lookUpInKmerTable(const char* s, len_t begin, len_t end) const {
        for (len_t i = begin; i < end; i++)
            if (s[i] == 'N') return RangePair();
        uint64_t key = 0;
        for (len_t i = 0; i < wordSize; i++) {
            int c = c2i(s[begin + i]);
            if (c < 1) return RangePair();
            key = (key << 2) | (uint64_t)(c - 1);
        }
        return kmerTable[key];
    }
};

typedef MatcherT<MoveIndexAdapter> MoveMatcher;

} // namespace orc

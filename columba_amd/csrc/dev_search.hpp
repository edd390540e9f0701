// Device-side search: partitioning, search-scheme DFS (edit + Hamming), task emission.
//
// One GPU thread walks one read x strand through the whole per-strand driver
//   SearchStrategy::matchWithSearches   reference src/searchstrategy.cpp:425-493
// i.e. partitioning (:141-419), part-level in-text pre-verification (:464-476), dynamic scheme
// selection (src/searchstrategy.h:2505-2537), doRecSearch (:1181-1254) and the recursive DFS
//   IndexInterface::recApproxMatchEdit / branchAndBound / goDeeper   src/indexinterface.cpp:377-669
//   IndexInterface::recApproxMatchHamming                           src/indexinterface.cpp:1211-1304
// The recursion over search phases is an explicit frame stack (one Frame per phase); the
// per-phase DFS stacks, matrix rows and cluster live in a per-thread scratch slab in HBM.
//
// What the reference does inline — locate (findSA), in-text verification, FM-occurrence
// conversion — is NOT done here: the DFS only emits compact work items into global queues
// (ballot-free atomic append), and dedicated regular kernels (kernels.hip) consume them.  The
// DFS never depends on the result of a verification, so the occurrence SET is unchanged.
#pragma once
#include "dev_index.hpp"
#include "dev_matrix.hpp"

namespace cmb {

constexpr int MAXP = 8;      // max parts (k <= 6 -> 7 parts in multiple_opt)
constexpr int MAXS = 16;     // max searches per scheme
constexpr int MAXSCH = 4;    // max alternative schemes per k (dynamic selection)
constexpr int MAX_READ = 256;
constexpr int CL_MAX = 32;   // cluster cells (Wh + Wv + 1)
constexpr int DESC_MAX = 56; // descendants handed to the next phase
constexpr int ROWS_MAX = MAX_READ + MAXP * 24;
constexpr int STACK_MAX = 3 * ROWS_MAX + 4 * MAXP;
constexpr int GW = (MAX_READ + 31) / 32 + 3;

// ---- strategy tables (built on the host by host/schemes.cpp) -------------------------------
struct DevSearch { // Search, src/search.h:55-101
    uint8_t n;
    uint8_t order[MAXP], L[MAXP], U[MAXP], dir[MAXP], dsw[MAXP];
    uint8_t low[MAXP], high[MAXP]; // lowestAndHighestPartsProcessedBefore[i]
    uint8_t uniAll, uniIdx;        // isUnidirectionalBackwards(i) = uniAll || i >= uniIdx (:479)
};
struct DevScheme {
    uint8_t nSearches, critical; // SearchScheme::criticalPartIndex (search.h:525)
    DevSearch s[MAXS];
};
struct DevStrategyK { // everything matchWithSearches needs for one distance k
    uint8_t metric, partition, numParts, nSchemes;
    uint32_t kmerCutOff;
    double seeding[MAXP]; // getSeedingPositions (searchstrategy.h:1825)
    uint64_t weights[MAXP]; // getWeights (:283)
    double begins[MAXP];  // getBegins (:245)
    DevScheme sch[MAXSCH];
};

// ---- work items -----------------------------------------------------------------------------
enum { ITEM_EDIT = 0, ITEM_HAMMING = 1, ITEM_EXACT = 2 };
// item = {rsId, saRow, startDiff | lengthBefore | remaining, meta}
// meta: shift[0:12) maxED[12:16) minED[16:20) fixed[20] kind[21:23)
__device__ __forceinline__ uint32_t packMeta(uint32_t shift, uint32_t maxED, uint32_t minED, uint32_t fixed,
                                             uint32_t kind) {
    return (shift & 0xFFFu) | (maxED << 12) | (minED << 16) | (fixed << 20) | (kind << 21);
}
struct FMOccRec { // in-index occurrence (FMOcc, src/indexhelpers.h:1353)
    uint32_t rsId, b, e, depth, dist, shift;
};
struct TextOccRec { // in-text occurrence before filtering
    uint32_t rsId, begin, end, dist;
};

enum { FLAG_ITEM_OVERFLOW = 1, FLAG_FMOCC_OVERFLOW = 2, FLAG_TEXT_OVERFLOW = 4, FLAG_CAPACITY = 8,
       FLAG_UNSUPPORTED_READ = 16, FLAG_DFS_OVERFLOW = 32 };

struct Queues {
    uint4* items;
    uint32_t itemCap;
    FMOccRec* fm;
    uint32_t fmCap;
    TextOccRec* text;
    uint32_t textCap;
    uint32_t* cnt; // [0] items, [1] fm, [2] text, [3] flags, [4] work counter, [5] dfs tasks, [6] dfs work counter
    unsigned long long* counters; // CMB_CNT_MAX
    uint32_t dbg;                 // development knobs (CMB_DEBUG), 0 in production
};

struct Node { // FMPosExt (src/indexhelpers.h:1544)
    RangePair r;
    uint16_t depth;
    uint8_t c; // 1..4 (A,C,G,T), 0 for the start cell
    uint8_t reported;
};

struct Frame { // one activation of recApproxMatchEdit / recApproxMatchHamming
    RangePair smR; // startMatch
    uint32_t smDist, smDepth, smShift;
    int8_t descLvl, otherLvl; // frames whose (desc, init) are `descendants` / `descOther`
    uint8_t idx, dir, uni, maxED, inReplay, useRev;
    uint16_t replay, xOff, xLen, rowBase, stackBase, stackTop;
    MatGeom g;
    // MatrixMetaInfo (src/indexhelpers.h:1677)
    uint8_t clSize;
    int8_t lastCell;
    uint16_t clEd[CL_MAX];
    Node clNode[CL_MAX];
    // descendants / initEds produced by goDeeper for the next phase (:615-636)
    uint8_t nDesc, nInit;
    Node desc[DESC_MAX];
    uint16_t init[DESC_MAX + 1];
};

struct Scratch {
    uint16_t pb[MAXP], pe[MAXP]; // parts (copied from k_partition's PartOut)
    uint64_t rowHP[ROWS_MAX], rowHN[ROWS_MAX], rowRAC[ROWS_MAX];
    uint16_t rowScore[ROWS_MAX];
    Node stack[STACK_MAX];
    Frame fr[MAXP];
};

struct Ctx {
    const DevIndex& ix;
    const DevStrategyK& st;
    Scratch& S;
    const Queues& q;
    uint32_t rsId, len, k;
    const uint8_t* seq; // read x strand as codes 1..4 (A,C,G,T), 5 = N   (k_prep)
    const uint32_t* G;  // match bit-strings [2][4][gw]: [0] forward read, [1] reversed read
    uint32_t gw;
    int dir;   // 0 FORWARD, 1 BACKWARD (definitions.h:103)
    bool uni;
    uint32_t cNode, cExp, cImm, cStart, cRows;
    uint32_t flags;
    __device__ Ctx(const DevIndex& i, const DevStrategyK& s, Scratch& sc, const Queues& qq)
        : ix(i), st(s), S(sc), q(qq), rsId(0), len(0), k(0), seq(nullptr), G(nullptr), gw(0), dir(1),
          uni(false), cNode(0), cExp(0), cImm(0), cStart(0), cRows(0), flags(0) {}
    __device__ __forceinline__ int mode() const { return uni ? 2 : (dir == 0 ? 0 : 1); }
    __device__ __forceinline__ void setDirection(int d, bool u) { // indexinterface.h:771-779
        dir = d;
        uni = u;
    }
};

// ---- queue appends --------------------------------------------------------------------------
__device__ __forceinline__ void emitItems(Ctx& c, const Range& sa, uint32_t a, uint32_t meta) {
    const uint32_t w = sa.width();
    if (!w) return;
    const uint32_t base = atomicAdd(&c.q.cnt[0], w);
    if (base + w > c.q.itemCap) {
        c.flags |= FLAG_ITEM_OVERFLOW;
        return;
    }
    for (uint32_t j = 0; j < w; j++) c.q.items[base + j] = make_uint4(c.rsId, sa.b + j, a, meta);
}
__device__ __forceinline__ void emitFMOcc(Ctx& c, const Range& sa, uint32_t depth, uint32_t dist, uint32_t shift) {
    const uint32_t base = atomicAdd(&c.q.cnt[1], 1u);
    if (base >= c.q.fmCap) {
        c.flags |= FLAG_FMOCC_OVERFLOW;
        return;
    }
    c.q.fm[base] = FMOccRec{c.rsId, sa.b, sa.e, depth, dist, shift};
}

// ---- extend helpers -------------------------------------------------------------------------
// IndexInterface::extendFMPos (indexinterface.cpp:675-697): push the non-empty children A,C,G,T
__device__ __forceinline__ void extendFMPos(Ctx& c, const RangePair& parent, uint32_t row, Frame& f) {
    uint32_t Rb[4], Re[4], db, de;
    const int md = c.mode();
    loadExtendRanks(c.ix, md, parent, Rb, Re, db, de);
    c.cExp++;
#pragma unroll
    for (uint32_t ch = 1; ch <= 4; ch++) {
        RangePair child;
        if (childFromRanks(c.ix, md, parent, ch, Rb, Re, db, de, child)) {
            if (f.stackTop >= STACK_MAX) {
                c.flags |= FLAG_CAPACITY;
                return;
            }
            Node& nd = c.S.stack[f.stackTop++];
            nd.r = child;
            nd.depth = (uint16_t)(row + 1);
            nd.c = (uint8_t)ch;
            nd.reported = 0;
            c.cNode++;
        }
    }
}

// character of part (b,e) with direction d at index i (Substring::operator[], substring.h:42,101)
__device__ __forceinline__ uint32_t partChar(const Ctx& c, uint32_t b, uint32_t e, int d, uint32_t i) {
    return d == 0 ? c.seq[b + i] : c.seq[e - i - 1];
}

// ---- cluster (MatrixMetaInfo) ---------------------------------------------------------------
struct OccTmp { // FMOcc under construction
    RangePair r;
    uint32_t dist, depth, shift;
    bool valid;
};
__device__ __forceinline__ void clSet(Frame& f, uint32_t idx, const Node& nd, uint32_t ed) { // :1723
    f.clEd[idx] = (uint16_t)ed;
    f.clNode[idx] = nd;
    f.clNode[idx].reported = 0;
    f.lastCell = (int8_t)idx;
}
// FMPosExt::report (indexhelpers.h:1586-1601)
__device__ __forceinline__ void nodeReport(Node& nd, OccTmp& m, uint32_t startDepth, uint32_t ed, bool once,
                                           uint32_t shift) {
    if (!nd.reported) {
        m.r = nd.r;
        m.dist = ed;
        m.depth = nd.depth + startDepth;
        m.shift = shift;
        m.valid = !nd.r.empty();
        if (once) nd.reported = 1;
    }
}

// ---- Hamming search (recApproxMatchHamming, indexinterface.cpp:1211-1304) -------------------
struct HammingSearch {
    Ctx& c;
    const DevSearch& s;
    __device__ HammingSearch(Ctx& cc, const DevSearch& ss) : c(cc), s(ss) {}
    __device__ __forceinline__ bool uniAt(int idx) const { return s.uniAll || idx >= (int)s.uniIdx; }

    __device__ void enter(int idx, int firstIdx, const RangePair& r, uint32_t dist, uint32_t depth) {
        Scratch& S = c.S;
        Frame& f = S.fr[idx];
        f.idx = (uint8_t)idx;
        f.smR = r;
        f.smDist = dist;
        f.smDepth = depth;
        f.dir = s.dir[idx];
        f.uni = uniAt(idx);
        f.maxED = s.U[idx];
        const int part = s.order[idx];
        f.xLen = (uint16_t)(S.pe[part] - S.pb[part]);
        f.rowBase = (uint16_t)(idx == firstIdx ? 0 : S.fr[idx - 1].rowBase + S.fr[idx - 1].xLen + 1);
        f.stackBase = (uint16_t)(idx == firstIdx ? 0 : S.fr[idx - 1].stackBase + 3 * (S.fr[idx - 1].xLen + 1) + 4);
        f.stackTop = f.stackBase;
        c.setDirection(f.dir, f.uni);
        S.rowScore[f.rowBase] = (uint16_t)dist;
        extendFMPos(c, r, 0, f);
    }

    __device__ void run(const RangePair& startR, uint32_t startDepth, int firstIdx) {
        Scratch& S = c.S;
        int level = firstIdx;
        enter(firstIdx, firstIdx, startR, 0, startDepth);
        const uint32_t sw = c.ix.switchPoint;
        const uint32_t maxEDs = s.U[s.n - 1], minEDs = s.L[s.n - 1];
        while (level >= firstIdx) {
            Frame& f = S.fr[level];
            c.setDirection(f.dir, f.uni);
            if (f.stackTop == f.stackBase) {
                level--;
                continue;
            }
            const Node nd = S.stack[--f.stackTop];
            const int idx = f.idx;
            const int part = s.order[idx];
            if (nd.r.width() <= sw) { // FMIndex::inTextVerificationHamming (fmindex.cpp:409-428)
                const uint32_t lengthBefore =
                    ((idx == 0) ? 0u : (uint32_t)S.pb[s.low[idx - 1]]) - (c.dir == 1 ? (uint32_t)nd.depth : 0u);
                emitItems(c, nd.r.sa, lengthBefore, packMeta(0, maxEDs, minEDs, 0, ITEM_HAMMING));
                continue;
            }
            const uint32_t row = nd.depth;
            const uint32_t pc = partChar(c, S.pb[part], S.pe[part], f.dir, row - 1);
            const uint32_t v = S.rowScore[f.rowBase + row - 1] + (nd.c != pc);
            S.rowScore[f.rowBase + row] = (uint16_t)v;
            if (v > f.maxED) continue;
            if (row == f.xLen) {
                if (v >= s.L[idx]) {
                    if (idx == s.n - 1) {
                        emitFMOcc(c, nd.r.sa, f.smDepth + f.xLen, v, 0);
                    } else {
                        enter(idx + 1, firstIdx, nd.r, v, f.smDepth + f.xLen);
                        level = idx + 1;
                    }
                }
                continue;
            }
            extendFMPos(c, nd.r, row, f);
        }
    }
};

} // namespace cmb

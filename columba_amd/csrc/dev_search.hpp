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

// getClusterCentra (indexhelpers.cpp:276-382) into f.desc / f.init
__device__ inline OccTmp clusterCentra(Ctx& c, Frame& f, uint32_t lowerBound) {
    OccTmp m;
    m.valid = false;
    m.dist = m.depth = m.shift = 0;
    m.r = RangePair{{0, 0}, {0, 0}};
    f.nDesc = 0;
    f.nInit = 0;
    const int last = f.lastCell;
    const uint32_t maxED = f.maxED;
    for (int i = 0; i <= last; i++) {
        if (f.clEd[i] > maxED || f.clEd[i] < lowerBound) continue;
        const bool betterThanParent = (i == 0) || f.clEd[i] <= f.clEd[i - 1];
        const bool betterThanChild = (i == last) || f.clEd[i] <= f.clEd[i + 1];
        if (!(betterThanParent && betterThanChild)) continue;
        nodeReport(f.clNode[i], m, f.smDepth, f.clEd[i], false, f.smShift);
        uint16_t* ie = f.init;
        uint32_t ni = 0;
        ie[ni++] = f.clEd[i];
        for (int j = i + 1; j <= last; j++) {
            f.desc[f.nDesc++] = f.clNode[j];
            ie[ni++] = f.clEd[j];
        }
        for (uint32_t kk = 1; kk < ni; kk++) {
            if (ie[kk] < lowerBound && ie[kk] <= ie[kk - 1] && (kk == ni - 1 || ie[kk] <= ie[kk + 1])) {
                uint32_t highestPoint = 0, lowestPoint = ni - 1;
                for (uint32_t l = kk; l-- > 0;) {
                    if (ie[l] != ie[l + 1] + 1) {
                        highestPoint = l + 1;
                        break;
                    }
                }
                for (uint32_t l = kk + 1; l < ni; l++) {
                    if (ie[l] != ie[l - 1] + 1) {
                        lowestPoint = l - 1;
                        break;
                    }
                }
                if (highestPoint != 0 && lowestPoint != ni - 1) {
                    uint32_t lC = lowestPoint, hC = highestPoint;
                    bool highest = true;
                    while (lC > hC) {
                        if (highest) {
                            ie[hC] = (uint16_t)min((int)maxED + 1, (int)ie[hC - 1] + 1);
                            hC++;
                        } else {
                            ie[lC] = (uint16_t)min((int)maxED + 1, (int)ie[lC + 1] + 1);
                            lC--;
                        }
                        highest = !highest;
                    }
                    if (lC == hC) ie[lC] = (uint16_t)min((int)ie[lC + 1] + 1, (int)ie[lC - 1] + 1);
                } else if (highestPoint == 0 && lowestPoint != ni - 1) {
                    for (uint32_t l = lowestPoint; l-- > 0;) ie[l] = (uint16_t)(ie[l + 1] + 1);
                } else if (highestPoint != 0 && lowestPoint == ni - 1) {
                    for (uint32_t l = highestPoint; l < ni; l++) ie[l] = (uint16_t)(ie[l - 1] + 1);
                }
            }
        }
        f.nInit = (uint8_t)ni;
        break;
    }
    return m;
}

// ---- edit-distance search over one Search (recApproxMatchEditEntry + recursion) -------------
struct EditSearch {
    Ctx& c;
    const DevSearch& s;
    int level;
    int firstIdx;
    __device__ EditSearch(Ctx& cc, const DevSearch& ss) : c(cc), s(ss), level(-1), firstIdx(0) {}

    __device__ __forceinline__ bool uniAt(int idx) const { return s.uniAll || idx >= (int)s.uniIdx; }

    // recApproxMatchEdit prologue (indexinterface.cpp:377-497)
    __device__ void enter(int idx, const OccTmp& sm, int prevLvl, int notPrevLvl) {
        Scratch& S = c.S;
        Frame& f = S.fr[idx];
        f.smR = sm.r;
        f.smDist = sm.dist;
        f.smDepth = sm.depth;
        f.smShift = sm.shift;
        f.idx = (uint8_t)idx;
        const int part = s.order[idx];
        f.maxED = s.U[idx];
        f.dir = s.dir[idx];
        const bool dsw = s.dsw[idx];
        f.descLvl = (int8_t)(dsw ? notPrevLvl : prevLvl);
        f.otherLvl = (int8_t)(dsw ? prevLvl : notPrevLvl);
        f.uni = uniAt(idx);
        c.setDirection(f.dir, f.uni);
        const uint32_t pb = S.pb[part], pe = S.pe[part];
        f.xLen = (uint16_t)(pe - pb);
        f.useRev = f.dir == 1;
        f.xOff = (uint16_t)(f.dir == 0 ? pb : c.len - pe);
        // first column of the band (:411-424)
        uint32_t initED[DESC_MAX + 1];
        uint32_t nInit;
        const Frame* df = f.descLvl >= 0 ? &S.fr[f.descLvl] : nullptr;
        const uint32_t nSrc = df ? df->nInit : 0;
        if (nSrc == 0) {
            initED[0] = sm.dist;
            nInit = 1;
        } else {
            uint32_t prevED = df->init[0];
            if (dsw)
                for (uint32_t i = 1; i < nSrc; i++) prevED = min(prevED, (uint32_t)df->init[i]);
            const uint32_t increase = sm.dist - prevED;
            for (uint32_t i = 0; i < nSrc; i++) initED[i] = df->init[i] + increase;
            nInit = nSrc;
        }
        uint64_t HP, HN, RAC;
        uint32_t score;
        initMatrix(f.g, f.xLen, f.maxED, initED, nInit, HP, HN, RAC, score);
        f.rowBase = (uint16_t)(idx == firstIdx ? 0 : S.fr[idx - 1].rowBase + S.fr[idx - 1].g.m);
        f.stackBase = (uint16_t)(idx == firstIdx ? 0 : S.fr[idx - 1].stackBase + 3 * S.fr[idx - 1].g.m + 4);
        f.stackTop = f.stackBase;
        f.nDesc = 0;
        f.nInit = 0;
        level = idx;
        if (f.g.Wv > 2 * MX_MAX_ED || f.g.sfc() > (uint32_t)CL_MAX || f.rowBase + f.g.m > (uint32_t)ROWS_MAX ||
            f.stackBase + 3 * f.g.m + 4 > (uint32_t)STACK_MAX) {
            c.flags |= FLAG_CAPACITY;
            f.inReplay = 0;
            return; // empty stack: the frame is left immediately
        }
        S.rowHP[f.rowBase] = HP;
        S.rowHN[f.rowBase] = HN;
        S.rowRAC[f.rowBase] = RAC;
        S.rowScore[f.rowBase] = (uint16_t)score;
        f.clSize = (uint8_t)f.g.sfc();
        f.lastCell = -1;
        for (uint32_t i = 0; i < f.clSize; i++) f.clEd[i] = (uint16_t)(f.maxED + 1);
        if (f.g.inFinalColumn(0)) { // :452-461
            Node nd;
            nd.r = sm.r;
            nd.depth = 0;
            nd.c = 0;
            nd.reported = 0;
            clSet(f, 0, nd, cellAt(0, f.xLen, HP, HN, score));
        }
        const uint32_t nDescSrc = df ? df->nDesc : 0;
        if (nDescSrc > 0) {
            f.inReplay = 1;
            f.replay = 0;
        } else {
            f.inReplay = 0;
            extendFMPos(c, sm.r, 0, f);
        }
    }

    // goToInTextVerificationEdit (indexinterface.cpp:340-375)
    __device__ void inTextSwitch(Frame& f, const Node& nd) {
        Scratch& S = c.S;
        const uint32_t st = S.pb[s.low[f.idx - 1]];
        const uint32_t maxEDs = s.U[s.n - 1], minEDs = s.L[s.n - 1];
        uint32_t startDiff = st + maxEDs;
        if (st == 0) {
            startDiff = 0;
        } else if (c.dir == 1) {
            const uint32_t row = nd.depth;
            const uint32_t col = f.g.firstColumn(row);
            const uint32_t ri = f.rowBase + row;
            startDiff -= col + cellAt(row, col, S.rowHP[ri], S.rowHN[ri], S.rowScore[ri]);
        } else if (f.otherLvl >= 0 && S.fr[f.otherLvl].nDesc > 0) {
            const Frame& o = S.fr[f.otherLvl];
            startDiff -= (uint32_t)o.nDesc - (uint32_t)o.nInit + (uint32_t)o.init[o.nInit - 1];
        }
        emitItems(c, nd.r.sa, startDiff, packMeta(f.smShift, maxEDs, minEDs, st == 0, ITEM_EDIT));
    }

    // goDeeper (indexinterface.cpp:563-669).  remFrom >= 0: remaining descendants start there.
    // Returns true if a deeper frame was entered.
    __device__ bool goDeeper(Frame& f, int remFrom) {
        Scratch& S = c.S;
        const int idx = f.idx;
        const int nIdx = idx + 1;
        const bool isEdge = s.order[idx] == 0 || s.order[idx] == s.n - 1;
        const uint32_t lowerBound = s.L[idx];
        if (isEdge) {
            if (nIdx == s.n) { // reportCentersAtEnd (indexhelpers.h:1743-1761)
                const int last = f.lastCell;
                for (int i = 0; i <= last; i++) {
                    if (f.clEd[i] <= f.maxED && (i == 0 || f.clEd[i] <= f.clEd[i - 1]) &&
                        (i == last || f.clEd[i] <= f.clEd[i + 1])) {
                        OccTmp m;
                        m.valid = false;
                        nodeReport(f.clNode[i], m, f.smDepth, f.clEd[i], true, f.smShift);
                        if (m.valid && m.dist >= lowerBound) emitFMOcc(c, m.r.sa, m.depth, m.dist, m.shift);
                    }
                }
                return false;
            }
            // reportDeepestMinimum (indexhelpers.h:1770-1798)
            uint32_t minED = f.maxED + 1;
            int hi = -1, deep = -1;
            for (int i = 0; i <= f.lastCell; i++) {
                if (f.clEd[i] < minED) {
                    minED = f.clEd[i];
                    hi = i;
                    deep = i;
                }
                if (f.clEd[i] == minED) deep = i;
            }
            OccTmp m;
            m.valid = false;
            if (minED <= f.maxED)
                nodeReport(f.clNode[deep], m, f.smDepth - (uint32_t)(deep - hi), minED, true,
                           ((c.dir == 1) ? (uint32_t)(deep - hi) : 0u) + f.smShift);
            if (m.valid && m.dist >= lowerBound) {
                enter(nIdx, m, -1, f.otherLvl);
                return true;
            }
            return false;
        }
        OccTmp nm = clusterCentra(c, f, lowerBound);
        if (!nm.valid) return false;
        if (remFrom >= 0) { // :625
            const Frame& df = S.fr[f.descLvl];
            for (int i = remFrom; i < (int)df.nDesc; i++) {
                if (f.nDesc >= DESC_MAX) {
                    c.flags |= FLAG_CAPACITY;
                    return false;
                }
                f.desc[f.nDesc++] = df.desc[i];
            }
        }
        for (uint32_t i = 0; i < f.nDesc; i++) f.desc[i].depth = (uint16_t)(i + 1); // :628
        const uint32_t maxEDNext = s.U[nIdx];
        while (f.init[f.nInit - 1] > maxEDNext) f.nInit--; // :634
        if (s.dsw[nIdx]) {
            if (f.nDesc > 0) {
                nm.r = f.desc[f.nDesc - 1].r;
                uint32_t mn = f.init[0];
                for (uint32_t i = 1; i < f.nInit; i++) mn = min(mn, (uint32_t)f.init[i]);
                nm.dist = mn;
            }
        }
        enter(nIdx, nm, idx, f.otherLvl);
        return true;
    }

    // branchAndBound (indexinterface.cpp:529-561): 0 = go on, 1 = prune, 2 = deeper frame entered
    __device__ int branchAndBound(Frame& f, const Node& nd, int remFrom) {
        Scratch& S = c.S;
        const uint32_t row = nd.depth;
        const uint32_t pi = f.rowBase + row - 1;
        uint64_t HP = S.rowHP[pi], HN = S.rowHN[pi], RAC = S.rowRAC[pi], D0;
        uint32_t score = S.rowScore[pi];
        const uint64_t M = matchWord(c.G + (f.useRev * 4 + (nd.c - 1)) * c.gw, f.xOff, f.xLen, row / MX_BLOCK);
        const bool valid = computeRow(f.g, row, M, HP, HN, D0, RAC, score);
        c.cRows++;
        S.rowHP[pi + 1] = HP;
        S.rowHN[pi + 1] = HN;
        S.rowRAC[pi + 1] = RAC;
        S.rowScore[pi + 1] = (uint16_t)score;
        if (f.g.inFinalColumn(row)) {
            const uint32_t clusterIdx = f.clSize + row - f.g.m;
            clSet(f, clusterIdx, nd, cellAt(row, f.g.n - 1, HP, HN, score));
            if (!valid || onlyVerticalGapsLeft(f.g, row, HN)) return goDeeper(f, remFrom) ? 2 : 1;
        }
        return valid ? 0 : 1;
    }

    // recApproxMatchEditEntry (indexinterface.cpp:1306-1325) + the whole recursion
    // (the width <= switch-point branch of the entry and SEARCH_STARTED are handled by k_partition)
    __device__ void run(const OccTmp& startMatch, int idx) {
        Scratch& S = c.S;
        firstIdx = idx;
        enter(idx, startMatch, -1, -1);
        const uint32_t sw = c.ix.switchPoint;
        while (level >= firstIdx) {
            Frame& f = S.fr[level];
            c.setDirection(f.dir, f.uni);
            if (f.inReplay) {
                const Frame& df = S.fr[f.descLvl];
                const uint32_t maxRow = f.g.m - 1;
                if (f.replay < df.nDesc && df.desc[f.replay].depth <= maxRow) {
                    const int i = f.replay++;
                    const int r = branchAndBound(f, df.desc[i], i + 1);
                    if (r == 2) continue;           // deeper frame runs; on its return this frame returns too
                    if (r == 1) leave();            // `return;` (:476)
                    continue;
                }
                if (df.desc[df.nDesc - 1].depth == maxRow) { // :479
                    leave();
                    continue;
                }
                const bool dsw = s.dsw[f.idx];
                const RangePair pair = dsw ? f.smR : df.desc[df.nDesc - 1].r;
                f.inReplay = 0;
                extendFMPos(c, pair, df.desc[df.nDesc - 1].depth, f);
                continue;
            }
            if (f.stackTop == f.stackBase) {
                leave();
                continue;
            }
            const Node nd = S.stack[--f.stackTop];
            const int r = branchAndBound(f, nd, -1);
            if (r != 0) continue; // pruned, or a deeper frame was entered
            if (nd.r.width() <= sw && f.idx != 0) {
                inTextSwitch(f, nd);
                continue;
            }
            extendFMPos(c, nd.r, nd.depth, f);
        }
    }
    // return from the current frame; a caller that was replaying descendants returns as well
    __device__ __forceinline__ void leave() {
        for (;;) {
            level--;
            if (level < firstIdx) return;
            if (!c.S.fr[level].inReplay) return;
        }
    }
};

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

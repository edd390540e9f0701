// Edit-distance DFS over one search (k_dfs_edit): wavefront-convergent formulation of
//   IndexInterface::recApproxMatchEdit / branchAndBound / goDeeper   reference src/indexinterface.cpp:377-669
//   MatrixMetaInfo                                                   reference src/indexhelpers.h:1677-1838, .cpp:276-382
//
// One lane owns one DfsTask.  The reference's DFS step — pop a node, compute its matrix row, push its
// children — is reorganised so that all 64 lanes of a wavefront execute the SAME two memory round
// trips per loop iteration, whatever their tasks are doing:
//   (1) EXPAND  : rank loads of the pending parent (12 x 16 B), then for its (<= 4) children the child
//                 range AND the child's matrix row (Hyyro/Myers update from the parent's row state,
//                 which is in registers) — the row state travels with the node on the stack, so the
//                 per-row matrix storage of the reference disappears;
//   (2) STEP    : pop one stack entry (48 B: ranges, depth, char, row state, validity) and classify it:
//                 prune / request expansion / stage an in-text verification / final-column handling.
// The rare, long paths (goDeeper with its cluster analysis, entering or leaving a phase, fetching a new
// task) are kept out of the two hot phases: a lane that needs one marks it (`pend`) and runs it in a
// fourth phase at the end of the same iteration, together with the other lanes that need one (parking
// lanes until more had gathered was measured slower).  Work items are appended with one atomic per
// wavefront (prefix sum).  Tasks are fetched in their emission order (sorting the wide start ranges
// first concentrated the heavy tasks in few wavefronts and was 2-3x slower).
// The frame header of the current phase lives in registers, the final-column edit distances of all
// phases in LDS; only nodes, descendants and suspended headers are in the per-task slab in HBM.
#pragma once
#include "dev_partition.hpp"

namespace cmb {

constexpr int STACK2_MAX = 3 * ROWS_MAX + 4 * MAXP;
constexpr int DON_MAX = 3; // children offered to helpers per expansion (the first valid child is always kept)

struct SEntry { // one DFS stack entry
    uint4 a;    // child ranges sa.b, sa.e, rev.b, rev.e
    uint4 b;    // x: depth | c << 16 | valid << 24 ; y: score ; z,w: HP
    uint4 c;    // x,y: HN ; z,w: RAC
};

struct Hot { // frame header of the running phase (registers)
    RangePair smR;
    uint32_t smDist, smDepth, smShift;
    MatGeom g;
    uint32_t xOff, xLen, stackBase, stackTop, replay;
    int descLvl, otherLvl;
    uint32_t idx, dir, uni, maxED, inReplay, useRev, clSize;
    int lastCell;
    uint64_t pHP, pHN, pRAC; // row state of the frame's current "parent row": row 0, then the last replayed row
    uint32_t pScore;
};

struct ColdFrame {
    Hot saved;             // header while the phase is suspended
    Node clNode[CL_MAX];   // MatrixMetaInfo::nodes
    uint8_t nDesc, nInit;  // descendants / initEds handed to the next phase (:615-636)
    Node desc[DESC_MAX];
    uint16_t init[DESC_MAX + 1];
};

struct Scratch2 {
    SEntry don[DON_MAX]; // children offered to helper lanes in this iteration
    SEntry subIn;        // child this lane received
    Hot donHdr;          // frame header of the phase the offered children belong to
    Hot subHdr;          // ... and of the phase of the child this lane received
    uint16_t pb[MAXP], pe[MAXP];
    ColdFrame fr[MAXP];
    SEntry stack[STACK2_MAX];
};

// The DFS task queue.  Positions [0, live): the first nStatic are assigned statically (lane l of wavefront
// w takes position l * W + w), the others through the counter q.cnt[6].  Positions map to tasks through
// order[] (widest start range first), so every wavefront starts with one task of each size class.
//
// Sharing a big subtree inside the wavefront: once the original tasks are exhausted, a lane without work
// becomes a helper.  A busy lane offers wide children (>= tSplit suffix-array rows, row not yet in the
// final column of its phase) to the helpers of its own wavefront instead of pushing them.  Such a child is
// self-contained given (i) its stack entry (ranges + row state), (ii) the header of its phase and (iii) the
// descendants / initial distances of the at most two earlier phases that header refers to (they do not
// change while the phase is alive); no cluster cell of the phase has been written on its path yet.  The
// entries and the header are staged in the donor's slab, matched to helpers by rank at the end of the
// iteration (ballots + one LDS table, no global atomics); the helper copies (i)-(iii) and runs the phase
// from that child as if it had popped it, with that phase as its bottom phase.  What no helper takes goes
// back on the donor's stack.  So the time a wavefront needs is the sum of its tasks over 64 lanes, not its
// largest task.
struct DfsQueue {
    const DfsTask* tasks;
    const uint32_t* order;
    uint32_t nStatic, live, tSplit;
};

enum { PEND_NONE = 0, PEND_FETCH, PEND_DEEPER, PEND_LEAVE };

struct EditDfs {
    const DevIndex& ix;
    const DevStrategyK& st;
    Scratch2& S;
    const Queues& q;
    uint8_t* clEd;                // LDS: [level][cell][lane], clCells cells per level
    uint32_t clCells;
    const uint32_t lane;
    // task
    const DevSearch* s = nullptr;
    uint32_t rsId = 0, len = 0, gw = 0;
    const uint32_t* G = nullptr;
    int level = -1, firstIdx = 0;
    uint32_t curTask = 0; // index of the running task (copied when a child is handed over)
    Hot H;
    // pending expansion
    bool req = false;
    RangePair reqParent;
    uint32_t reqRow = 0, reqScore = 0;
    uint64_t reqHP = 0, reqHN = 0, reqRAC = 0;
    // cached match words of one 32-row block of the running phase
    uint32_t mbBlock = 0xFFFFFFFFu;
    uint64_t Mblk[4];
    // parked heavy operation
    int pend = PEND_FETCH;
    bool firstFetch = true;
    bool origDone = false;            // the original tasks of this pass are exhausted
    uint32_t claim = 0xFFFFFFFFu;     // task index of the subtree this lane received (entry in S.subIn)
    uint32_t nDon = 0;                // children staged in S.don[] this iteration
    int pendRem = -1;
    // staged in-text work item
    uint32_t stN = 0, stB = 0, stA = 0, stMeta = 0;
    // counters
    uint32_t cNode = 0, cExp = 0, cRows = 0, flags = 0;

    __device__ EditDfs(const DevIndex& i, const DevStrategyK& t, Scratch2& sc, const Queues& qq,
                       uint8_t* ed, uint32_t cells, uint32_t ln)
        : ix(i), st(t), S(sc), q(qq), clEd(ed), clCells(cells), lane(ln) {}

    __device__ __forceinline__ bool uniAt(int idx) const { return s->uniAll || idx >= (int)s->uniIdx; }
    __device__ __forceinline__ int mode() const { return H.uni ? 2 : (H.dir == 0 ? 0 : 1); }
    __device__ __forceinline__ uint8_t& ED(int lvl, int cell) {
        return clEd[((uint32_t)lvl * clCells + (uint32_t)cell) * 64u + lane];
    }
    __device__ __forceinline__ const uint32_t* gbits(uint32_t ch) const { return G + (H.useRev * 4 + ch) * gw; }

    __device__ __forceinline__ void requestExpand(const RangePair& parent, uint32_t row, uint64_t HP, uint64_t HN,
                                                  uint64_t RAC, uint32_t score) {
        req = true;
        reqParent = parent;
        reqRow = row;
        reqHP = HP;
        reqHN = HN;
        reqRAC = RAC;
        reqScore = score;
    }

    // ---- (1) EXPAND: extendFMPos (indexinterface.cpp:675-697) + the children's rows (computeRow :536)
    __device__ __forceinline__ void expand(const DfsQueue& dq, bool helpersWaiting) {
        uint32_t Rb[4], Re[4], db, de;
        const int md = mode();
        loadExtendRanks(ix, md, reqParent, Rb, Re, db, de);
        cExp++;
        const uint32_t row = reqRow + 1;
        const uint32_t blk = row / MX_BLOCK;
        if (blk != mbBlock) {
            mbBlock = blk;
#pragma unroll
            for (int ch = 0; ch < 4; ch++) Mblk[ch] = matchWord(gbits(ch), H.xOff, H.xLen, blk);
        }
        const bool canSplit = helpersWaiting && dq.tSplit != 0 && !H.g.inFinalColumn(row);
        bool kept = false;
#pragma unroll
        for (uint32_t ch = 1; ch <= 4; ch++) {
            RangePair child;
            if (childFromRanks(ix, md, reqParent, ch, Rb, Re, db, de, child)) {
                uint64_t HP = reqHP, HN = reqHN, RAC = reqRAC, D0;
                uint32_t score = reqScore;
                const bool valid = computeRow(H.g, row, Mblk[ch - 1], HP, HN, D0, RAC, score);
                cNode++;
                cRows++;
                // a child whose row already exceeds maxED outside the final column is pruned the moment it
                // is popped (branchAndBound returns true, :560) and has no other effect: do not push it
                if (!valid && !H.g.inFinalColumn(row)) continue;
                SEntry e;
                e.a = make_uint4(child.sa.b, child.sa.e, child.rev.b, child.rev.e);
                e.b = make_uint4(row | (ch << 16) | ((valid ? 1u : 0u) << 24), score, (uint32_t)HP, (uint32_t)(HP >> 32));
                e.c = make_uint4((uint32_t)HN, (uint32_t)(HN >> 32), (uint32_t)RAC, (uint32_t)(RAC >> 32));
                if (canSplit && kept && child.sa.width() >= dq.tSplit) { // offer the subtree to a helper
                    if (nDon == 0) S.donHdr = H;
                    S.don[nDon++] = e;
                    continue;
                }
                kept = true;
                if (H.stackTop >= (uint32_t)STACK2_MAX) {
                    flags |= FLAG_CAPACITY;
                    break;
                }
                S.stack[H.stackTop++] = e;
            }
        }
        req = false;
    }

    // ---- cluster helpers (MatrixMetaInfo) -----------------------------------------------------
    __device__ __forceinline__ void clSet(uint32_t cell, const Node& nd, uint32_t ed) { // indexhelpers.h:1723
        ED(level, cell) = (uint8_t)min(ed, 255u);
        Node n2 = nd;
        n2.reported = 0;
        S.fr[level].clNode[cell] = n2;
        H.lastCell = (int)cell;
    }

    // getClusterCentra (indexhelpers.cpp:276-382) into the cold frame's desc / init
    __device__ __forceinline__ OccTmp clusterCentra(uint32_t lowerBound) {
        ColdFrame& f = S.fr[level];
        OccTmp m;
        m.valid = false;
        m.dist = m.depth = m.shift = 0;
        m.r = RangePair{{0, 0}, {0, 0}};
        f.nDesc = 0;
        f.nInit = 0;
        const int last = H.lastCell;
        const uint32_t maxED = H.maxED;
        for (int i = 0; i <= last; i++) {
            const uint32_t ei = ED(level, i);
            if (ei > maxED || ei < lowerBound) continue;
            const bool betterThanParent = (i == 0) || ei <= ED(level, i - 1);
            const bool betterThanChild = (i == last) || ei <= ED(level, i + 1);
            if (!(betterThanParent && betterThanChild)) continue;
            Node ci = f.clNode[i];
            nodeReport(ci, m, H.smDepth, ei, false, H.smShift);
            uint16_t* ie = f.init;
            uint32_t ni = 0;
            ie[ni++] = (uint16_t)ei;
            uint32_t nd = 0;
            for (int j = i + 1; j <= last; j++) {
                f.desc[nd++] = f.clNode[j];
                ie[ni++] = ED(level, j);
            }
            f.nDesc = (uint8_t)nd;
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

    __device__ __forceinline__ void emitFM(const Range& sa, uint32_t depth, uint32_t dist, uint32_t shift) {
        const uint32_t base = atomicAdd(&q.cnt[1], 1u);
        if (base >= q.fmCap) {
            flags |= FLAG_FMOCC_OVERFLOW;
            return;
        }
        q.fm[base] = FMOccRec{rsId, sa.b, sa.e, depth, dist, shift};
    }

    // ---- phase entry: recApproxMatchEdit prologue (indexinterface.cpp:377-497) ------------------
    // Called with `level` = the caller's level (or firstIdx - 1 for the first phase); suspends the caller.
    __device__ __forceinline__ void enter(int idx, const OccTmp& sm, int prevLvl, int notPrevLvl) {
        uint32_t parentStackEnd = 0;
        if (idx != firstIdx) {
            S.fr[level].saved = H;
            parentStackEnd = H.stackBase + 3 * H.g.m + 4;
        }
        level = idx;
        mbBlock = 0xFFFFFFFFu;
        ColdFrame& f = S.fr[idx];
        H.smR = sm.r;
        H.smDist = sm.dist;
        H.smDepth = sm.depth;
        H.smShift = sm.shift;
        H.idx = (uint32_t)idx;
        const int part = s->order[idx];
        H.maxED = s->U[idx];
        H.dir = s->dir[idx];
        const bool dsw = s->dsw[idx];
        H.descLvl = dsw ? notPrevLvl : prevLvl;
        H.otherLvl = dsw ? prevLvl : notPrevLvl;
        H.uni = uniAt(idx) ? 1u : 0u;
        const uint32_t pb = S.pb[part], pe = S.pe[part];
        H.xLen = pe - pb;
        H.useRev = H.dir == 1 ? 1u : 0u;
        H.xOff = H.dir == 0 ? pb : len - pe;
        // first column of the band (:411-424)
        const ColdFrame* df = H.descLvl >= 0 ? &S.fr[H.descLvl] : nullptr;
        const uint32_t nSrc = df ? df->nInit : 0;
        uint32_t first = sm.dist, last = sm.dist, nInit = 1, increase = 0;
        if (nSrc != 0) {
            uint32_t prevED = df->init[0];
            if (dsw)
                for (uint32_t i = 1; i < nSrc; i++) prevED = min(prevED, (uint32_t)df->init[i]);
            increase = sm.dist - prevED;
            first = df->init[0] + increase;
            last = df->init[nSrc - 1] + increase;
            nInit = nSrc;
        }
        initMatrix(H.g, H.xLen, H.maxED, first, last, df ? df->init : nullptr, increase, nInit, H.pHP, H.pHN, H.pRAC,
                   H.pScore);
        H.stackBase = parentStackEnd;
        H.stackTop = H.stackBase;
        H.replay = 0;
        H.inReplay = 0;
        H.lastCell = -1;
        H.clSize = H.g.sfc();
        f.nDesc = 0;
        f.nInit = 0;
        if (H.g.Wv > 2 * MX_MAX_ED || H.clSize > clCells || H.stackBase + 3 * H.g.m + 4 > (uint32_t)STACK2_MAX) {
            flags |= FLAG_CAPACITY;
            H.clSize = 0;
            return; // empty stack: the phase is left at its first step
        }
        for (uint32_t i = 0; i < H.clSize; i++) ED(idx, i) = (uint8_t)(H.maxED + 1);
        if (H.g.inFinalColumn(0)) { // :452-461
            Node nd;
            nd.r = sm.r;
            nd.depth = 0;
            nd.c = 0;
            nd.reported = 0;
            clSet(0, nd, cellAt(0, H.xLen, H.pHP, H.pHN, H.pScore));
        }
        const uint32_t nDescSrc = df ? df->nDesc : 0;
        if (nDescSrc > 0) {
            H.inReplay = 1;
        } else {
            requestExpand(sm.r, 0, H.pHP, H.pHN, H.pRAC, H.pScore);
        }
    }

    // return from the running phase; a caller that was replaying descendants returns as well (:472-477)
    __device__ __forceinline__ void leave() {
        for (;;) {
            level--;
            if (level < firstIdx) return;
            if (!S.fr[level].saved.inReplay) break;
        }
        H = S.fr[level].saved;
        mbBlock = 0xFFFFFFFFu;
    }

    // a phase entry decided by heavy()'s branches, performed at ONE call site (enter() is large)
    struct EnterReq {
        bool want;
        int idx, prevLvl, notPrevLvl;
        OccTmp sm;
    };

    // goDeeper (indexinterface.cpp:563-669).  Returns true if a deeper phase is to be entered (er filled).
    __device__ __forceinline__ bool goDeeper(int remFrom, EnterReq& er) {
        ColdFrame& f = S.fr[level];
        const int idx = (int)H.idx;
        const int nIdx = idx + 1;
        const bool isEdge = s->order[idx] == 0 || s->order[idx] == s->n - 1;
        const uint32_t lowerBound = s->L[idx];
        if (isEdge) {
            if (nIdx == s->n) { // reportCentersAtEnd (indexhelpers.h:1743-1761)
                const int last = H.lastCell;
                for (int i = 0; i <= last; i++) {
                    const uint32_t ei = ED(level, i);
                    if (ei <= H.maxED && (i == 0 || ei <= ED(level, i - 1)) && (i == last || ei <= ED(level, i + 1))) {
                        OccTmp m;
                        m.valid = false;
                        Node ci = f.clNode[i];
                        if (!ci.reported) {
                            nodeReport(ci, m, H.smDepth, ei, true, H.smShift);
                            f.clNode[i].reported = 1;
                        }
                        if (m.valid && m.dist >= lowerBound) emitFM(m.r.sa, m.depth, m.dist, m.shift);
                    }
                }
                return false;
            }
            // reportDeepestMinimum (indexhelpers.h:1770-1798)
            uint32_t minED = H.maxED + 1;
            int hi = -1, deep = -1;
            for (int i = 0; i <= H.lastCell; i++) {
                const uint32_t ei = ED(level, i);
                if (ei < minED) {
                    minED = ei;
                    hi = i;
                    deep = i;
                }
                if (ei == minED) deep = i;
            }
            OccTmp m;
            m.valid = false;
            if (minED <= H.maxED) {
                Node cd = f.clNode[deep];
                if (!cd.reported) {
                    nodeReport(cd, m, H.smDepth - (uint32_t)(deep - hi), minED, true,
                               ((H.dir == 1) ? (uint32_t)(deep - hi) : 0u) + H.smShift);
                    f.clNode[deep].reported = 1;
                }
            }
            if (m.valid && m.dist >= lowerBound) {
                er = EnterReq{true, nIdx, -1, H.otherLvl, m};
                return true;
            }
            return false;
        }
        OccTmp nm = clusterCentra(lowerBound);
        if (!nm.valid) return false;
        if (remFrom >= 0) { // :625
            const ColdFrame& df = S.fr[H.descLvl];
            for (int i = remFrom; i < (int)df.nDesc; i++) {
                if (f.nDesc >= DESC_MAX) {
                    flags |= FLAG_CAPACITY;
                    return false;
                }
                f.desc[f.nDesc++] = df.desc[i];
            }
        }
        for (uint32_t i = 0; i < f.nDesc; i++) f.desc[i].depth = (uint16_t)(i + 1); // :628
        const uint32_t maxEDNext = s->U[nIdx];
        while (f.init[f.nInit - 1] > maxEDNext) f.nInit--; // :634
        if (s->dsw[nIdx]) {
            if (f.nDesc > 0) {
                nm.r = f.desc[f.nDesc - 1].r;
                uint32_t mn = f.init[0];
                for (uint32_t i = 1; i < f.nInit; i++) mn = min(mn, (uint32_t)f.init[i]);
                nm.dist = mn;
            }
        }
        er = EnterReq{true, nIdx, idx, H.otherLvl, nm};
        return true;
    }

    // ---- classification of one node whose row is known (branchAndBound + the loop body :506-526) --
    __device__ __forceinline__ void classify(const Node& nd, uint64_t HP, uint64_t HN, uint64_t RAC, uint32_t score,
                                             bool valid, int remFrom) {
        const uint32_t row = nd.depth;
        if (H.g.inFinalColumn(row)) {
            const uint32_t clusterIdx = H.clSize + row - H.g.m;
            clSet(clusterIdx, nd, cellAt(row, H.g.n - 1, HP, HN, score));
            if (!valid || onlyVerticalGapsLeft(H.g, row, HN)) {
                pend = PEND_DEEPER;
                pendRem = remFrom;
                return;
            }
        }
        if (!valid) {
            if (H.inReplay) pend = PEND_LEAVE; // `return;` (:476)
            return;
        }
        if (H.inReplay) return; // next descendant
        if (nd.r.width() <= ix.switchPoint && H.idx != 0) { // goToInTextVerificationEdit (:340-375)
            const uint32_t stt = S.pb[s->low[H.idx - 1]];
            const uint32_t maxEDs = s->U[s->n - 1], minEDs = s->L[s->n - 1];
            uint32_t startDiff = stt + maxEDs;
            if (stt == 0) {
                startDiff = 0;
            } else if (H.dir == 1) {
                const uint32_t col = H.g.firstColumn(row);
                startDiff -= col + cellAt(row, col, HP, HN, score);
            } else if (H.otherLvl >= 0 && S.fr[H.otherLvl].nDesc > 0) {
                const ColdFrame& o = S.fr[H.otherLvl];
                startDiff -= (uint32_t)o.nDesc - (uint32_t)o.nInit + (uint32_t)o.init[o.nInit - 1];
            }
            stN = nd.r.sa.width();
            stB = nd.r.sa.b;
            stA = startDiff;
            stMeta = packMeta(H.smShift, maxEDs, minEDs, stt == 0, ITEM_EDIT);
            return;
        }
        requestExpand(nd.r, row, HP, HN, RAC, score);
    }

    // ---- (2) STEP -------------------------------------------------------------------------------
    __device__ __forceinline__ void step() {
        if (H.inReplay) {
            const ColdFrame& df = S.fr[H.descLvl];
            const uint32_t maxRow = H.g.m - 1;
            if (H.replay < df.nDesc && df.desc[H.replay].depth <= maxRow) {
                const int i = (int)H.replay++;
                const Node nd = df.desc[i];
                const uint64_t M = matchWord(gbits(nd.c - 1), H.xOff, H.xLen, nd.depth / MX_BLOCK);
                uint64_t D0;
                const bool valid = computeRow(H.g, nd.depth, M, H.pHP, H.pHN, D0, H.pRAC, H.pScore);
                cRows++;
                classify(nd, H.pHP, H.pHN, H.pRAC, H.pScore, valid, i + 1);
                return;
            }
            const Node lastD = df.desc[df.nDesc - 1];
            if (lastD.depth == maxRow) { // :479
                pend = PEND_LEAVE;
                return;
            }
            RangePair pair = lastD.r; // (no `?:` between two objects: a select of addresses would pin
            if (s->dsw[H.idx]) pair = H.smR; //  the whole lane state in scratch memory)
            H.inReplay = 0;
            requestExpand(pair, lastD.depth, H.pHP, H.pHN, H.pRAC, H.pScore);
            return;
        }
        if (H.stackTop == H.stackBase) {
            pend = PEND_LEAVE;
            return;
        }
        const SEntry e = S.stack[--H.stackTop];
        Node nd;
        nd.r = RangePair{{e.a.x, e.a.y}, {e.a.z, e.a.w}};
        nd.depth = (uint16_t)(e.b.x & 0xFFFFu);
        nd.c = (uint8_t)((e.b.x >> 16) & 0xFFu);
        nd.reported = 0;
        const bool valid = (e.b.x >> 24) & 1u;
        const uint64_t HP = (uint64_t)e.b.z | ((uint64_t)e.b.w << 32);
        const uint64_t HN = (uint64_t)e.c.x | ((uint64_t)e.c.y << 32);
        const uint64_t RAC = (uint64_t)e.c.z | ((uint64_t)e.c.w << 32);
        classify(nd, HP, HN, RAC, e.b.y, valid, -1);
    }

    // helper side of a hand-over: copy the child, the header of its phase and the two earlier frames that
    // header refers to (descendants / initial distances only) from the donor's slab
    __device__ __forceinline__ void receive(const Scratch2& D, uint32_t item) {
        S.subIn = D.don[item];
        const Hot h = D.donHdr;
        S.subHdr = h;
        const int refs[2] = {h.descLvl, h.otherLvl};
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int lv = refs[r];
            if (lv < 0 || (r == 1 && lv == refs[0])) continue;
            const ColdFrame& src = D.fr[lv];
            ColdFrame& dst = S.fr[lv];
            const uint32_t nd = src.nDesc, ni = src.nInit;
            dst.nDesc = (uint8_t)nd;
            dst.nInit = (uint8_t)ni;
            for (uint32_t i = 0; i < nd; i++) dst.desc[i] = src.desc[i];
            for (uint32_t i = 0; i < ni; i++) dst.init[i] = src.init[i];
        }
    }

    // ---- (4) parked operations ------------------------------------------------------------------
    __device__ __forceinline__ void heavy(const DfsQueue& dq, const PartOut* parts, const uint64_t* offs,
                                          const uint32_t* Gall) {
        EnterReq er;
        er.want = false;
        bool subStart = false;
        if (pend == PEND_FETCH) {
            uint32_t p = 0xFFFFFFFFu;
            if (claim != 0xFFFFFFFFu) { // a subtree received from a lane of this wavefront
                curTask = claim;
                claim = 0xFFFFFFFFu;
                subStart = true;
                p = 0;
            } else {
                if (firstFetch) {
                    firstFetch = false;
                    const uint32_t g = (threadIdx.x & 63u) * gridDim.x + blockIdx.x;
                    if (g < dq.nStatic) p = g;
                }
                if (p == 0xFFFFFFFFu) p = atomicAdd(&q.cnt[6], 1u);
                if (p >= dq.live) {
                    p = 0xFFFFFFFFu;
                    origDone = true; // stays PEND_FETCH: from now on the lane is a helper
                    level = -1;
                    firstIdx = 0;
                } else {
                    curTask = dq.order[p];
                }
            }
            DfsTask task;
            if (p != 0xFFFFFFFFu) {
                task = dq.tasks[curTask];
                if (task.rsId == 0xFFFFFFFFu) p = 0xFFFFFFFFu; // a hole of the task queue: try again
            }
            if (p != 0xFFFFFFFFu) {
                pend = PEND_NONE;
                rsId = task.rsId;
                len = (uint32_t)(offs[(rsId >> 1) + 1] - offs[rsId >> 1]);
                G = Gall + (size_t)rsId * 8 * gw;
                const PartOut po = parts[rsId];
#pragma unroll
                for (int i = 0; i < MAXP; i++) {
                    S.pb[i] = po.pb[i];
                    S.pe[i] = po.pe[i];
                }
                s = &st.sch[task.scheme].s[task.search];
                req = false;
                if (subStart) {
                    // continue the donor's phase from the received child: that phase is this lane's bottom phase
                    H = S.subHdr;
                    level = (int)H.idx;
                    firstIdx = level;
                    mbBlock = 0xFFFFFFFFu;
                    H.stackBase = 0;
                    H.stackTop = 0;
                    H.replay = 0;
                    H.inReplay = 0;
                    H.lastCell = -1;
                    for (uint32_t i = 0; i < H.clSize; i++) ED(level, i) = (uint8_t)(H.maxED + 1);
                    S.fr[level].nDesc = 0;
                    S.fr[level].nInit = 0;
                    S.stack[H.stackTop++] = S.subIn;
                } else {
                    firstIdx = task.idx;
                    level = firstIdx - 1;
                    er.want = true;
                    er.idx = firstIdx;
                    er.prevLvl = -1;
                    er.notPrevLvl = -1;
                    er.sm.r = task.r;
                    er.sm.dist = 0;
                    er.sm.depth = task.depth;
                    er.sm.shift = 0;
                    er.sm.valid = true;
                }
            }
        } else {
            if (pend == PEND_DEEPER) {
                pend = PEND_NONE;
                const bool wasReplay = H.inReplay;
                const bool entered = goDeeper(pendRem, er);
                if (!entered && wasReplay) pend = PEND_LEAVE; // branchAndBound returned true inside the replay (:472-477)
            }
            if (pend == PEND_LEAVE) {
                pend = PEND_NONE;
                leave();
                if (level < firstIdx) pend = PEND_FETCH; // search finished: next task
            }
        }
        if (er.want) enter(er.idx, er.sm, er.prevLvl, er.notPrevLvl);
    }
};

} // namespace cmb

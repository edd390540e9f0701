// Edit-distance search of one search scheme as a FRONTIER of independent nodes (k_bfs_*):
//   IndexInterface::recApproxMatchEdit / branchAndBound / goDeeper   reference src/indexinterface.cpp:377-669
//   MatrixMetaInfo                                                   reference src/indexhelpers.h:1677-1838, .cpp:276-382
//
// The reference walks the search tree of one read depth first on one thread.  Nothing in that walk needs
// the depth-first ORDER: a node's matrix row depends only on its parent's row, and the only state shared
// between sibling paths — the "reported" mark of a final-column node (indexhelpers.h:1586-1601) — is a
// test-and-set whose winner does not matter (every contender would report the same occurrence).  So the
// device keeps ALL live nodes of ALL reads in one frontier in HBM and advances it level by level:
//
//   k_bfs_expand  one lane per frontier node (64 B, read coalesced): the two rank blocks of the node, the hot
//                 32 B of its phase context and the four match words of its row block are fetched in ONE
//                 round trip; the lane computes the <= 4 children (extendFMPos + computeRow), classifies them
//                 (branchAndBound, in-text switch) and appends nodes / events / in-text items with ONE
//                 atomic per queue and 256 nodes (block-wide prefix sums).  No stack, no per-task state.
//   k_bfs_heavy   one lane per EVENT — a path that ended in the final column of its phase (goDeeper), or a
//                 task that starts its first approximate phase: cluster analysis, occurrence reports,
//                 creation of the next phase's context, replay of the handed-over descendants.
//
// Immutable records replace the reference's mutable per-thread state:
//   context  (384 B)  one activation of recApproxMatchEdit: band geometry, start match, in-text switch
//                     parameters, the match words of the part (8 blocks x 4 nucleotides) and references to
//                     the contexts that hold the `descendants` / `descOther` lists it was entered with;
//   F record (32 B)   a node in the final column of its phase: ranges, depth, character, link to the
//                     final-column node above it on its path.  The chain of F records of a path, with the
//                     edit distances of its final-column cells (5 bits per cell, carried by the node and
//                     handed to the event), IS its MatrixMetaInfo; only the event handler ever walks it.
#pragma once
// (included by kernels.hpp after its wave helpers: waveExclusiveScan)
#include "dev_partition.hpp"

namespace cmb {

constexpr uint32_t BFS_NONE = 0xFFFFFFFFu;
constexpr uint32_t ED_CELLS = 24; // final-column cells per phase (5 bits each in a 128-bit pack); 3k+2 <= 24 for k <= 7
// A context is three 128-byte lines: line 0 the cold header (C0..C4), line 1 the hot word and the match words of
// row blocks 0..2 — what an expansion reads of its context is ONE line for the first 96 rows of a phase (with the
// hot word in line 0 the kernel took 72 instead of 66 ms) —, line 2 the match words of row blocks 3..6.
constexpr uint32_t CTX_U4 = 24;
constexpr uint32_t CTX_HOT = 8;   // uint4 index of the hot word
constexpr uint32_t CTX_M = 10;    // uint4 index of the match words of row block 0 ({A,C}, {G,T} per block)
constexpr uint32_t CTX_MBLK = 7;  // row blocks with cached match words (rows < 224)
constexpr uint32_t F_U4 = 2;      // uint4 per F record: {ranges} {depth | c << 16, parent, reported, -}
// (all blocks of a pass should be resident together — 3 blocks of 256 threads per CU at ~160 VGPRs — or the event
// blocks, which come last in the grid, only start when expansion blocks have finished)
constexpr uint32_t BFS_GRID = 576;    // blocks that expand the frontier (grid-stride)
constexpr uint32_t BFS_GRID_EV = 192; // blocks that handle the events of the same pass

// (the FLAG_BFS_* bits live in dev_search.hpp with all other bits of the flag word)
// any of these set by an earlier pass: the frontier is incomplete, later passes do nothing (the host re-runs)
constexpr uint32_t BFS_STOP = FLAG_BFS_Q | FLAG_BFS_EV | FLAG_BFS_F | FLAG_BFS_CTX | FLAG_BFS_ARENA | FLAG_ITEM_OVERFLOW |
                              FLAG_FMOCC_OVERFLOW | FLAG_CAPACITY;

// "has an earlier pass stopped the search?" as ONE answer per block: other blocks of the same launch may set the
// flag word while this one starts, so thread 0 reads it once and the block branches on the LDS copy (a per-thread
// read could let some wavefronts leave before a barrier the others wait at).
__device__ __forceinline__ bool blockStopped(const Queues& q) {
    __shared__ uint32_t stopWord;
    if (threadIdx.x == 0) stopWord = __hip_atomic_load(&q.cnt[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & BFS_STOP;
    __syncthreads();
    return stopWord != 0u;
}

struct BfsBufs {
    uint4* Q[2];  // frontier nodes, 4 planes of qCap: {ranges} {row | score << 16, ctx, fc, RAC bit} {HP, HN}
                  // {final-column distances of the path: only touched for nodes in the final column}
    uint4* Ev[2]; // events, 2 x 16 B: {ctx, F index of the node that ended its path, remaining-descendants index | -1,
                  // cell} {final-column distances of the path}
    uint4* F;     // final-column records, 2 x 16 B: {ranges} {depth | c << 16, parent, reported, -}
    uint4* C;     // contexts, CTX_U4 x 16 B
    uint4* A;     // list arena: descendants (2 x 16 B each: ranges, {depth | c << 16}) and initial distances (u16)
    uint32_t qCap, evCap, fCap, cCap, aCap;
    uint32_t* nq;    // [pass] number of frontier nodes consumed by pass `pass`
    uint32_t* ne;    // [pass] number of events consumed by pass `pass`
    uint32_t* pool;  // [0] F records, [1] contexts, [2] arena units handed out
    unsigned long long* blockCnt; // [BFS_GRID][4] per-block counters: nodes, expansions, rows, -
};

struct EdPack { // final-column edit distances of one path, cell i at bits [5i, 5i+5)
    uint64_t lo, hi;
};
__device__ __forceinline__ uint32_t edGet(const EdPack& p, uint32_t i) {
    return i < 12u ? (uint32_t)(p.lo >> (5u * i)) & 31u : (uint32_t)(p.hi >> (5u * (i - 12u))) & 31u;
}
__device__ __forceinline__ void edPut(EdPack& p, uint32_t i, uint32_t v) { // cell i is still zero
    if (i < 12u) p.lo |= (uint64_t)v << (5u * i);
    else p.hi |= (uint64_t)v << (5u * (i - 12u));
}

__device__ __forceinline__ uint64_t u64of(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }

// block-wide exclusive prefix sum + ONE atomic for the whole block.  Every thread of the (256-thread) block
// calls; returns this thread's first slot.  `sh` = 5 words of LDS per call site in flight.
__device__ __forceinline__ uint32_t blockAppend(uint32_t* counter, uint32_t n, uint32_t* sh, uint32_t& blockTotal) {
    uint32_t wTotal;
    const uint32_t pre = waveExclusiveScan(n, wTotal);
    const uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) sh[w] = wTotal;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = sh[0] + sh[1] + sh[2] + sh[3];
        sh[4] = t ? atomicAdd(counter, t) : 0u;
    }
    __syncthreads();
    uint32_t off = sh[4];
    blockTotal = sh[0] + sh[1] + sh[2] + sh[3];
    if (w > 0) off += sh[0];
    if (w > 1) off += sh[1];
    if (w > 2) off += sh[2];
    __syncthreads(); // sh may be reused by the next call
    return off + pre;
}

// Four appends at once: the four wave scans first, ONE barrier, four lanes issue the four atomics side by side
// (one atomic round trip per tile instead of four), one more barrier.  sh = 4 x 5 words.
__device__ __forceinline__ void blockAppend4(uint32_t* c0, uint32_t* c1, uint32_t* c2, uint32_t* c3, const uint32_t n[4],
                                             uint32_t (*sh)[5], uint32_t off[4]) {
    uint32_t pre[4];
    const uint32_t w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t wTotal;
        pre[j] = waveExclusiveScan(n[j], wTotal);
        if ((threadIdx.x & 63u) == 0) sh[j][w] = wTotal;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint32_t* c = threadIdx.x == 0 ? c0 : threadIdx.x == 1 ? c1 : threadIdx.x == 2 ? c2 : c3;
        const uint32_t t = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
        sh[threadIdx.x][4] = t ? atomicAdd(c, t) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t o = sh[j][4] + pre[j];
        if (w > 0) o += sh[j][0];
        if (w > 1) o += sh[j][1];
        if (w > 2) o += sh[j][2];
        off[j] = o;
    }
    __syncthreads(); // sh may be reused by the next call
}

// ------------------------------------------------------------------ expand
// One lane per frontier node.  Children (extendFMPos, indexinterface.cpp:675-697) get their matrix row at
// once (computeRow; the reference computes it when the child is popped — every pushed child is popped) and
// are classified as in branchAndBound (:529-561) + the stack loop (:506-526):
//   row invalid outside the final column      -> dropped
//   final column, row invalid or only vertical gaps left -> event (goDeeper)
//   narrow range (in-text switch, :516)        -> in-text verification items
//   otherwise                                  -> node of the next frontier
__device__ __forceinline__ void bfsExpand(const DevIndex& ix, const BfsBufs& B, uint32_t pass, const Queues& q,
                                          uint32_t bid, uint32_t nBlocks) {
    __shared__ uint32_t sh[4][5];
    const uint32_t nIn = min(B.nq[pass], B.qCap);
    const uint4* __restrict__ Qi = B.Q[pass & 1u];
    uint4* __restrict__ Qo = B.Q[(pass + 1u) & 1u];
    uint4* __restrict__ Eo = B.Ev[(pass + 1u) & 1u];
    const uint32_t qCap = B.qCap;
    uint32_t cNode = 0, cExp = 0, cRows = 0, flags = 0;
    for (uint32_t base = bid * 256u; base < nIn; base += nBlocks * 256u) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        const bool act = i < nIn;
        // per child: 0 nothing, 1 node, 2 event, 3 in-text items
        uint32_t kinds = 0; // 4 bits per child
        uint4 cr[4];                       // child ranges
        uint32_t cw[4];                    // child: score
        uint64_t cHP[4], cHN[4], cRAC[4];  // child row state
        uint32_t cEd[4];                   // child: final-column edit distance (needF) / in-text start difference
        uint32_t needF = 0;                // bit ch: the child is in the final column and gets an F record
        uint32_t row1 = 0, ctx = 0, fcP = BFS_NONE, rsId = 0, itMeta = 0, cell = 0;
        EdPack pack{0, 0};
        if (act) {
            const uint4 n0 = Qi[i], n1 = Qi[(size_t)qCap + i], n2 = Qi[(size_t)2 * qCap + i];
            ctx = n1.y;
            fcP = n1.z;
            const uint32_t row = n1.x & 0xFFFFu, score = n1.x >> 16;
            row1 = row + 1;
            const uint4* Cx = B.C + (size_t)ctx * CTX_U4;
            const uint32_t blk = row1 / MX_BLOCK;
            // ---- the single memory step: context (hot part), match words, F pack, rank blocks
            const uint4 hot = Cx[CTX_HOT]; // everything the expansion needs of its context, in ONE 16-byte request
            const uint4 mA = Cx[CTX_M + 2 * blk], mB = Cx[CTX_M + 1 + 2 * blk];
            uint4 fp = make_uint4(0, 0, 0, 0);
            if (fcP != BFS_NONE) fp = Qi[(size_t)3 * qCap + i]; // final-column distances of the path so far
            const uint32_t dir = (hot.x >> 27) & 1u, uni = (hot.x >> 28) & 1u;
            const int md = uni ? 2 : (dir == 0 ? 0 : 1);
            const RangePair parent{{n0.x, n0.y}, {n0.z, n0.w}};
            uint32_t Rb[4], Re[4], db, de;
            loadExtendRanks(ix, md, parent, Rb, Re, db, de);
            cExp++;
            rsId = hot.x & 0x1FFFFFFu;
            itMeta = hot.w & 0x7FFFFFu;
            MatGeom g;
            g.n = hot.y & 0x1FFu;
            g.m = (hot.y >> 9) & 0x1FFu;
            g.Wv = (hot.y >> 18) & 31u;
            g.Wh = (hot.y >> 23) & 15u;
            g.maxED = (hot.y >> 27) & 15u;
            const uint32_t clSize = hot.w >> 23;
            const uint32_t itMode = (hot.x >> 25) & 3u; // 0: phase 0 (no switch), 1: start difference fixed, 2: BACKWARD
            pack = EdPack{u64of(fp.x, fp.y), u64of(fp.z, fp.w)};
            const uint64_t pHP = u64of(n2.x, n2.y), pHN = u64of(n2.z, n2.w), pRAC = 1ull << (n1.w & 63u); // (RAC is always a single bit)
            const bool inFC = g.inFinalColumn(row1);
            cell = clSize + row1 - g.m;
            if (inFC && cell >= ED_CELLS) { // (a row beyond the matrix: cannot happen for a well-formed phase)
                flags |= FLAG_CAPACITY;
                cell = ED_CELLS - 1;
            }
            const uint64_t Mw[4] = {u64of(mA.x, mA.y), u64of(mA.z, mA.w), u64of(mB.x, mB.y), u64of(mB.z, mB.w)};
#pragma unroll
            for (uint32_t ch = 1; ch <= 4; ch++) {
                RangePair child;
                if (!childFromRanks(ix, md, parent, ch, Rb, Re, db, de, child)) continue;
                cNode++;
                cRows++;
                uint64_t HP = pHP, HN = pHN, RAC = pRAC, D0;
                uint32_t sc = score;
                const bool valid = computeRow(g, row1, Mw[ch - 1], HP, HN, D0, RAC, sc);
                if (!valid && !inFC) continue; // pruned when popped (branchAndBound returns true, :560)
                cr[ch - 1] = make_uint4(child.sa.b, child.sa.e, child.rev.b, child.rev.e);
                cw[ch - 1] = sc;
                cHP[ch - 1] = HP;
                cHN[ch - 1] = HN;
                cRAC[ch - 1] = RAC;
                if (inFC) {
                    const uint32_t ed = cellAt(row1, g.n - 1, HP, HN, sc);
                    cEd[ch - 1] = min(ed, 31u);
                    if (ed > 31u) flags |= FLAG_CAPACITY;
                    if (!valid || onlyVerticalGapsLeft(g, row1, HN)) {
                        kinds |= 2u << (4 * (ch - 1));
                        needF |= 1u << (ch - 1);
                        continue;
                    }
                }
                if (child.sa.width() <= ix.switchPoint && itMode != 0) { // goToInTextVerificationEdit (:340-375)
                    uint32_t startDiff = hot.z;
                    if (itMode == 2) {
                        const uint32_t col = g.firstColumn(row1);
                        startDiff -= col + cellAt(row1, col, HP, HN, sc);
                    }
                    cEd[ch - 1] = startDiff; // (a child that leaves the index needs no F record)
                    kinds |= 3u << (4 * (ch - 1));
                    continue;
                }
                kinds |= 1u << (4 * (ch - 1));
                if (inFC) needF |= 1u << (ch - 1);
            }
        }
        // ---- block-wide allocation in the four output queues
        uint32_t nNode = 0, nEv = 0, nIt = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t kd = (kinds >> (4 * c)) & 15u;
            nNode += kd == 1;
            nEv += kd == 2;
            if (kd == 3) nIt += cr[c].y - cr[c].x;
        }
        const uint32_t nF = (uint32_t)__popc(needF);
        const uint32_t want[4] = {nNode, nEv, nIt, nF};
        uint32_t got[4];
        blockAppend4(&B.nq[pass + 1], &B.ne[pass + 1], &q.cnt[0], &B.pool[0], want, sh, got);
        uint32_t oNode = got[0], oEv = got[1], oIt = got[2], oF = got[3];
        // (a block whose share does not fit drops it: the host sees the needed sizes and re-runs)
        bool okNode = true, okEv = true, okIt = true, okF = true;
        if (oNode + nNode > qCap) { okNode = false; flags |= FLAG_BFS_Q; }
        if (oEv + nEv > B.evCap) { okEv = false; flags |= FLAG_BFS_EV; }
        if (oIt + nIt > q.itemCap) { okIt = false; flags |= FLAG_ITEM_OVERFLOW; }
        if (oF + nF > B.fCap) { okF = false; flags |= FLAG_BFS_F; }
        if (act && okF && okNode && okEv && okIt) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t kd = (kinds >> (4 * c)) & 15u;
                if (kd == 0) continue;
                uint32_t fc = BFS_NONE;
                if (needF & (1u << c)) {
                    fc = oF++;
                    EdPack p2 = pack;
                    edPut(p2, cell, cEd[c]);
                    uint4* Fr = B.F + (size_t)fc * F_U4;
                    Fr[0] = cr[c];
                    Fr[1] = make_uint4(row1 | ((uint32_t)(c + 1) << 16), fcP, 0u, 0u);
                }
                if (kd == 1) {
                    const uint32_t o = oNode++;
                    Qo[o] = cr[c];
                    Qo[(size_t)qCap + o] = make_uint4(row1 | (cw[c] << 16), ctx, fc, (uint32_t)__ffsll((unsigned long long)cRAC[c]) - 1u);
                    Qo[(size_t)2 * qCap + o] = make_uint4((uint32_t)cHP[c], (uint32_t)(cHP[c] >> 32), (uint32_t)cHN[c],
                                                          (uint32_t)(cHN[c] >> 32));
                    if (needF & (1u << c)) {
                        EdPack p2 = pack;
                        edPut(p2, cell, cEd[c]);
                        Qo[(size_t)3 * qCap + o] = make_uint4((uint32_t)p2.lo, (uint32_t)(p2.lo >> 32), (uint32_t)p2.hi, (uint32_t)(p2.hi >> 32));
                    }
                } else if (kd == 2) {
                    {
                        EdPack p2 = pack;
                        edPut(p2, cell, cEd[c]);
                        Eo[(size_t)2 * oEv] = make_uint4(ctx, fc, 0xFFFFFFFFu, cell);
                        Eo[(size_t)2 * oEv + 1] = make_uint4((uint32_t)p2.lo, (uint32_t)(p2.lo >> 32), (uint32_t)p2.hi, (uint32_t)(p2.hi >> 32));
                        oEv++;
                    }
                } else {
                    const uint32_t w = cr[c].y - cr[c].x;
                    for (uint32_t t = 0; t < w; t++) q.items[oIt + t] = make_uint4(rsId, cr[c].x + t, cEd[c], itMeta);
                    oIt += w;
                }
            }
        }
    }
    // per-block counters (summed by k_bfs_finish): one writer per slot and launch, launches are ordered
    unsigned long long v[3] = {cNode, cExp, cRows};
#pragma unroll
    for (int j = 0; j < 3; j++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[j] += __shfl_xor(v[j], d);
    }
    __shared__ unsigned long long shc[4][3];
    if ((threadIdx.x & 63u) == 0)
        for (int j = 0; j < 3; j++) shc[threadIdx.x >> 6][j] = v[j];
    __syncthreads();
    if (threadIdx.x < 3) {
        const unsigned long long t = shc[0][threadIdx.x] + shc[1][threadIdx.x] + shc[2][threadIdx.x] + shc[3][threadIdx.x];
        if (t) B.blockCnt[(size_t)bid * 4 + threadIdx.x] += t;
    }
    if (flags) atomicOr(&q.cnt[3], flags);
}

__global__ void k_bfs_finish(BfsBufs B, Queues q) { // one block: per-block counters -> the batch counters
    __shared__ unsigned long long s[3];
    if (threadIdx.x < 3) s[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < BFS_GRID * 4; j += blockDim.x) {
        const unsigned long long v = B.blockCnt[j];
        if ((j & 3u) < 3u && v) atomicAdd(&s[j & 3u], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&q.counters[0], s[0]);  // NODE_COUNTER
        atomicAdd(&q.counters[7], s[1]);  // EXPANSIONS
        atomicAdd(&q.counters[12], s[1]); // DFS_EXPANSIONS
        atomicAdd(&q.counters[11], s[2]); // MATRIX_ROWS
    }
}

// ------------------------------------------------------------------ events: goDeeper + phase entry
struct HeavyPlan {
    uint32_t kind;      // 0 nothing, 1 reportCentersAtEnd, 2 reportDeepestMinimum + entry, 3 getClusterCentra + entry,
                        // 4 first phase of a task
    uint32_t centres;   // kind 1: mask of the cluster centres
    uint32_t ci;        // kind 3: cell of the centre; kind 2: cell of the deepest minimum
    uint32_t hi;        // kind 2: highest cell holding the minimum
    uint32_t ed;        // kind 2/3: edit distance of that cell
    uint32_t nRem;      // kind 3: descendants of the interrupted replay that are appended
    uint32_t nDescNew, ni; // kind 3: sizes of the lists handed to the next phase
    uint32_t nDescSrc;  // descendants the next phase replays
};

template <bool START>
__device__ __forceinline__ void bfsHeavy(const DevIndex& ix, const DevStrategyK* __restrict__ stp, const BfsBufs& B,
                                         uint32_t pass, const DfsTask* __restrict__ tasks, uint32_t nTasks,
                                         const uint64_t* __restrict__ offs, uint32_t gw, const uint32_t* __restrict__ G,
                                         const PartOut* __restrict__ parts, const Queues& q, uint32_t bid,
                                         uint32_t nBlocks) {
    __shared__ uint32_t sh[4][5];
    __shared__ uint8_t ieL[ED_CELLS + 2][256]; // initEds under construction, [entry][thread]
    const uint32_t outP = START ? 0u : pass + 1u;
    const uint32_t nIn = START ? nTasks : min(B.ne[pass], B.evCap);
    const uint4* __restrict__ Ei = B.Ev[pass & 1u];
    uint4* __restrict__ Qo = B.Q[outP & 1u];
    uint4* __restrict__ Eo = B.Ev[outP & 1u];
    const uint32_t qCap = B.qCap;
    const uint32_t tid = threadIdx.x;
    uint32_t cRows = 0, flags = 0;
    for (uint32_t base = bid * 256u; base < nIn; base += nBlocks * 256u) { // block-uniform trip count
        const uint32_t i = base + tid;
        HeavyPlan P{};
        // the context the event belongs to (START: none)
        uint32_t rsId = 0, scheme = 0, search = 0, idx = 0, dirCur = 0, smDepth = 0, smShift = 0, maxED = 0;
        uint32_t fcE = BFS_NONE, last = 0, c0i = BFS_NONE, descRef0 = BFS_NONE, otherRef0 = BFS_NONE, lowerBound = 0;
        int remFrom = -1;
        EdPack pack{0, 0};
        RangePair startR{{0, 0}, {0, 0}};
        uint32_t startDepth = 0;
        const DevSearch* s = nullptr;
        if (i < nIn) {
            if (START) {
                const DfsTask t = tasks[i];
                if (t.rsId != 0xFFFFFFFFu) { // (holes of the task queue)
                    P.kind = 4;
                    rsId = t.rsId;
                    scheme = t.scheme;
                    search = t.search;
                    idx = t.idx; // the phase to enter
                    startR = t.r;
                    startDepth = t.depth;
                    s = &stp->sch[scheme].s[search];
                }
            } else {
                const uint4 ev = Ei[(size_t)2 * i];
                c0i = ev.x;
                fcE = ev.y;
                remFrom = (int)ev.z;
                last = ev.w;
                const uint4* Cx = B.C + (size_t)c0i * CTX_U4;
                const uint4 c0 = Cx[0], c1 = Cx[1], c3 = Cx[3];
                const uint4 fp = Ei[(size_t)2 * i + 1]; // final-column distances of the path (travel with the event)
                pack = EdPack{u64of(fp.x, fp.y), u64of(fp.z, fp.w)};
                rsId = c0.x;
                maxED = (c0.z >> 16) & 0xFFu;
                const uint32_t fl = c0.w;
                idx = fl & 15u;
                dirCur = (fl >> 4) & 1u;
                scheme = (fl >> 12) & 15u;
                search = (fl >> 16) & 31u;
                descRef0 = c1.z;
                otherRef0 = c1.w;
                smDepth = c3.x;
                smShift = c3.y;
                s = &stp->sch[scheme].s[search];
                const uint32_t nIdx = idx + 1;
                const bool isEdge = s->order[idx] == 0 || s->order[idx] == s->n - 1;
                lowerBound = s->L[idx];
                if (isEdge) {
                    if (nIdx == s->n) { // reportCentersAtEnd (indexhelpers.h:1743-1761)
                        uint32_t m = 0;
                        for (uint32_t c = 0; c <= last; c++) {
                            const uint32_t e = edGet(pack, c);
                            if (e <= maxED && (c == 0 || e <= edGet(pack, c - 1)) && (c == last || e <= edGet(pack, c + 1)))
                                m |= 1u << c;
                        }
                        P.centres = m;
                        P.kind = m ? 1u : 0u;
                    } else { // reportDeepestMinimum (indexhelpers.h:1770-1798)
                        uint32_t minED = maxED + 1, hi = 0, deep = 0;
                        for (uint32_t c = 0; c <= last; c++) {
                            const uint32_t e = edGet(pack, c);
                            if (e < minED) {
                                minED = e;
                                hi = c;
                                deep = c;
                            }
                            if (e == minED) deep = c;
                        }
                        if (minED <= maxED) {
                            P.kind = 2;
                            P.ci = deep;
                            P.hi = hi;
                            P.ed = minED;
                        }
                    }
                } else { // getClusterCentra (indexhelpers.cpp:276-382): the first centre at or above the lower bound
                    for (uint32_t c = 0; c <= last; c++) {
                        const uint32_t e = edGet(pack, c);
                        if (e > maxED || e < lowerBound) continue;
                        if ((c == 0 || e <= edGet(pack, c - 1)) && (c == last || e <= edGet(pack, c + 1))) {
                            P.kind = 3;
                            P.ci = c;
                            P.ed = e;
                            break;
                        }
                    }
                    if (P.kind == 3) {
                        const uint32_t nd0 = last - P.ci;
                        P.ni = nd0 + 1;
                        if (remFrom >= 0) { // :625 (descRef0 is valid: the event came out of a replay)
                            const uint32_t dn = B.C[(size_t)descRef0 * CTX_U4 + 4].y & 0xFFu;
                            P.nRem = dn > (uint32_t)remFrom ? dn - (uint32_t)remFrom : 0u;
                        }
                        P.nDescNew = nd0 + P.nRem;
                        if (P.nDescNew > (uint32_t)DESC_MAX) {
                            flags |= FLAG_CAPACITY;
                            P.kind = 0;
                        }
                    }
                }
            }
        }
        // ---- what the entry of the next phase will need
        uint32_t idxN = 0, descRefN = BFS_NONE, otherRefN = BFS_NONE;
        bool descSelf = false, otherSelf = false;
        uint4 dC4 = make_uint4(0, 0, 0, 0); // list header of the context holding `descendants`
        if (P.kind >= 2) {
            idxN = P.kind == 4 ? idx : idx + 1;
            const bool dswN = s->dsw[idxN];
            // prevDir lists: produced by this event (kind 3) or none; notPrevDir lists: the event's `descOther`
            const bool prevSelf = P.kind == 3;
            const uint32_t notPrev = P.kind == 4 ? BFS_NONE : otherRef0;
            if (dswN) {
                descRefN = notPrev;
                otherSelf = prevSelf;
            } else {
                descSelf = prevSelf;
                otherRefN = notPrev;
            }
            if (descSelf) P.nDescSrc = P.nDescNew;
            else if (descRefN != BFS_NONE) {
                dC4 = B.C[(size_t)descRefN * CTX_U4 + 4];
                P.nDescSrc = dC4.y & 0xFFu;
            }
        }
        // ---- block-wide allocation: contexts, F records, arena units, in-index occurrences
        const uint32_t wantCtx = P.kind >= 2 ? 1u : 0u;
        const uint32_t wantF = P.kind >= 2 ? 1u + P.nDescSrc : 0u;
        const uint32_t wantA = P.kind == 3 ? 2u * P.nDescNew + (2u * P.ni + 15u) / 16u : 0u;
        const uint32_t wantFm = P.kind == 1 ? (uint32_t)__popc(P.centres) : 0u;
        const uint32_t want[4] = {wantCtx, wantF, wantA, wantFm};
        uint32_t got[4];
        blockAppend4(&B.pool[1], &B.pool[0], &B.pool[2], &q.cnt[1], want, sh, got);
        const uint32_t cNew = got[0], aOff = got[2];
        uint32_t fNext = got[1], fmNext = got[3];
        bool ok = true;
        if (cNew + wantCtx > B.cCap) { ok = false; flags |= FLAG_BFS_CTX; }
        if (fNext + wantF > B.fCap) { ok = false; flags |= FLAG_BFS_F; }
        if (aOff + wantA > B.aCap) { ok = false; flags |= FLAG_BFS_ARENA; }
        if (fmNext + wantFm > q.fmCap) { ok = false; flags |= FLAG_FMOCC_OVERFLOW; }
        if (!ok) P.kind = 0;

        // ---- the event itself
        bool enter = false;
        RangePair smR{{0, 0}, {0, 0}};
        uint32_t smDist = 0, smDepthN = 0, smShiftN = 0;
        uint32_t nInitNew = 0;
        if (P.kind == 4) {
            enter = true;
            smR = startR;
            smDepthN = startDepth;
        } else if (P.kind == 1) {
            // walk the path's final-column chain from the last cell up to the highest centre
            const uint32_t fmEnd = fmNext + wantFm;
            uint32_t cur = fcE;
            const uint32_t lowest = (uint32_t)__ffs(P.centres) - 1u;
            for (uint32_t c = last;; c--) {
                const uint4 f1 = B.F[(size_t)cur * F_U4 + 1];
                if ((P.centres >> c) & 1u) {
                    const uint32_t old = atomicExch(&reinterpret_cast<uint32_t*>(B.F + (size_t)cur * F_U4 + 1)[2], 1u);
                    if (!old) { // FMPosExt::report (indexhelpers.h:1586-1601): once per node
                        const uint4 r = B.F[(size_t)cur * F_U4];
                        const uint32_t e = edGet(pack, c);
                        if (r.y > r.x && e >= lowerBound)
                            q.fm[fmNext++] = FMOccRec{rsId, r.x, r.y, (f1.x & 0xFFFFu) + smDepth, e, smShift};
                    }
                }
                if (c == lowest) break;
                cur = f1.y;
            }
            for (; fmNext < fmEnd; fmNext++) q.fm[fmNext].rsId = 0xFFFFFFFFu; // holes
        } else if (P.kind == 2) {
            uint32_t cur = fcE;
            for (uint32_t c = last; c > P.ci; c--) cur = B.F[(size_t)cur * F_U4 + 1].y;
            const uint32_t old = atomicExch(&reinterpret_cast<uint32_t*>(B.F + (size_t)cur * F_U4 + 1)[2], 1u);
            if (!old) {
                const uint4 r = B.F[(size_t)cur * F_U4];
                const uint4 f1 = B.F[(size_t)cur * F_U4 + 1];
                const uint32_t up = P.ci - P.hi;
                smR = RangePair{{r.x, r.y}, {r.z, r.w}};
                smDist = P.ed;
                smDepthN = (f1.x & 0xFFFFu) + (smDepth - up);
                smShiftN = (dirCur == 1 ? up : 0u) + smShift;
                enter = !smR.empty() && smDist >= lowerBound;
            }
        } else if (P.kind == 3) {
            // descendants = the final-column nodes below the centre (walked bottom-up), then the rest of the
            // interrupted replay; depths renumbered 1.. (:627-630)
            uint4* dl = B.A + aOff;
            uint32_t cur = fcE;
            for (uint32_t c = last; c > P.ci; c--) {
                const uint4 r = B.F[(size_t)cur * F_U4];
                const uint4 f1 = B.F[(size_t)cur * F_U4 + 1];
                const uint32_t j = c - P.ci - 1;
                dl[2 * j] = r;
                dl[2 * j + 1] = make_uint4((j + 1) | (f1.x & 0xFF0000u), 0u, 0u, 0u);
                cur = f1.y;
            }
            const uint4 r = B.F[(size_t)cur * F_U4];
            const uint4 f1 = B.F[(size_t)cur * F_U4 + 1];
            smR = RangePair{{r.x, r.y}, {r.z, r.w}};
            smDist = P.ed;
            smDepthN = (f1.x & 0xFFFFu) + smDepth;
            smShiftN = smShift;
            enter = !smR.empty();
            if (enter) {
                const uint32_t nd0 = last - P.ci;
                if (P.nRem) {
                    const uint4 sC4 = B.C[(size_t)descRef0 * CTX_U4 + 4];
                    const uint4* sl = B.A + sC4.x;
                    for (uint32_t t = 0; t < P.nRem; t++) {
                        const uint32_t j = nd0 + t;
                        dl[2 * j] = sl[2 * ((uint32_t)remFrom + t)];
                        dl[2 * j + 1] = make_uint4((j + 1) | (sl[2 * ((uint32_t)remFrom + t) + 1].x & 0xFF0000u), 0u, 0u, 0u);
                    }
                }
                // initEds (indexhelpers.cpp:300-376): the centre's distance, the distances below it, then the
                // rewrite of clusters that dip under the lower bound
                const uint32_t ni = P.ni;
                for (uint32_t j = 0; j < ni; j++) ieL[j][tid] = (uint8_t)edGet(pack, P.ci + j);
                for (uint32_t kk = 1; kk < ni; kk++) {
                    const uint32_t ek = ieL[kk][tid];
                    if (ek < lowerBound && ek <= ieL[kk - 1][tid] && (kk == ni - 1 || ek <= ieL[kk + 1][tid])) {
                        uint32_t highestPoint = 0, lowestPoint = ni - 1;
                        for (uint32_t l = kk; l-- > 0;) {
                            if ((uint32_t)ieL[l][tid] != (uint32_t)ieL[l + 1][tid] + 1u) {
                                highestPoint = l + 1;
                                break;
                            }
                        }
                        for (uint32_t l = kk + 1; l < ni; l++) {
                            if ((uint32_t)ieL[l][tid] != (uint32_t)ieL[l - 1][tid] + 1u) {
                                lowestPoint = l - 1;
                                break;
                            }
                        }
                        if (highestPoint != 0 && lowestPoint != ni - 1) {
                            uint32_t lC = lowestPoint, hC = highestPoint;
                            bool highest = true;
                            while (lC > hC) {
                                if (highest) {
                                    ieL[hC][tid] = (uint8_t)min((int)maxED + 1, (int)ieL[hC - 1][tid] + 1);
                                    hC++;
                                } else {
                                    ieL[lC][tid] = (uint8_t)min((int)maxED + 1, (int)ieL[lC + 1][tid] + 1);
                                    lC--;
                                }
                                highest = !highest;
                            }
                            if (lC == hC) ieL[lC][tid] = (uint8_t)min((int)ieL[lC + 1][tid] + 1, (int)ieL[lC - 1][tid] + 1);
                        } else if (highestPoint == 0 && lowestPoint != ni - 1) {
                            for (uint32_t l = lowestPoint; l-- > 0;) ieL[l][tid] = (uint8_t)(ieL[l + 1][tid] + 1);
                        } else if (highestPoint != 0 && lowestPoint == ni - 1) {
                            for (uint32_t l = highestPoint; l < ni; l++) ieL[l][tid] = (uint8_t)(ieL[l - 1][tid] + 1);
                        }
                    }
                }
                nInitNew = ni;
                const uint32_t maxEDNext = s->U[idxN];
                while (nInitNew > 1 && ieL[nInitNew - 1][tid] > maxEDNext) nInitNew--; // :634
                uint16_t* il = reinterpret_cast<uint16_t*>(dl + 2 * P.nDescNew);
                uint32_t mn = ieL[0][tid];
                for (uint32_t j = 0; j < nInitNew; j++) {
                    il[j] = ieL[j][tid];
                    mn = min(mn, (uint32_t)ieL[j][tid]);
                }
                if (s->dsw[idxN] && P.nDescNew > 0) { // :640-648
                    const uint4 lr = dl[2 * (P.nDescNew - 1)];
                    smR = RangePair{{lr.x, lr.y}, {lr.z, lr.w}};
                    smDist = mn;
                }
            }
        }

        // ---- phase entry: recApproxMatchEdit prologue + replay of the descendants (:377-497)
        uint32_t outKind = 0; // 1: node of the next frontier, 2: event
        uint4 oN0 = make_uint4(0, 0, 0, 0), oN1 = oN0, oN2 = oN0, oN4 = oN0, oEv = oN0, oEv1 = oN0;
        if (enter) {
            if (descSelf) descRefN = cNew;
            if (otherSelf) otherRefN = cNew;
            const uint32_t part = s->order[idxN];
            const uint32_t maxEDn = s->U[idxN];
            const uint32_t dirN = s->dir[idxN];
            const bool dswN = s->dsw[idxN];
            const uint32_t uniN = (s->uniAll || idxN >= s->uniIdx) ? 1u : 0u;
            const PartOut po = parts[rsId];
            const uint32_t len = (uint32_t)(offs[(rsId >> 1) + 1] - offs[rsId >> 1]);
            const uint32_t pb = po.pb[part], pe = po.pe[part];
            const uint32_t xLen = pe - pb;
            const uint32_t useRev = dirN == 1 ? 1u : 0u;
            const uint32_t xOff = dirN == 0 ? pb : len - pe;
            // lists this phase was entered with
            uint32_t dListOff = 0, nSrcDesc = 0, nSrcInit = 0;
            if (descSelf) {
                dListOff = aOff;
                nSrcDesc = P.nDescNew;
                nSrcInit = nInitNew;
            } else if (descRefN != BFS_NONE) {
                dListOff = dC4.x;
                nSrcDesc = dC4.y & 0xFFu;
                nSrcInit = (dC4.y >> 8) & 0xFFu;
            }
            const uint4* dl = B.A + dListOff;
            const uint16_t* il = reinterpret_cast<const uint16_t*>(dl + 2 * nSrcDesc);
            uint32_t first = smDist, lastI = smDist, nInit = 1, increase = 0;
            if (nSrcInit != 0) { // :411-424
                uint32_t prevED = il[0];
                if (dswN)
                    for (uint32_t j = 1; j < nSrcInit; j++) prevED = min(prevED, (uint32_t)il[j]);
                increase = smDist - prevED;
                first = il[0] + increase;
                lastI = il[nSrcInit - 1] + increase;
                nInit = nSrcInit;
            }
            MatGeom g;
            uint64_t HP, HN, RAC;
            uint32_t score;
            initMatrix(g, xLen, maxEDn, first, lastI, nSrcInit ? il : nullptr, increase, nInit, HP, HN, RAC, score);
            const uint32_t clSize = g.sfc();
            const uint32_t nBlk = (g.m - 1) / MX_BLOCK + 1;
            if (g.Wv > 2 * MX_MAX_ED || clSize > ED_CELLS || nBlk > CTX_MBLK || g.m > 0xFFFFu) {
                flags |= FLAG_CAPACITY;
            } else {
                // in-text switch parameters of the phase (goToInTextVerificationEdit, :340-375)
                uint32_t itMode = 0, itStart = 0, itMeta = 0;
                if (idxN != 0) {
                    const uint32_t stt = po.pb[s->low[idxN - 1]];
                    const uint32_t maxEDs = s->U[s->n - 1], minEDs = s->L[s->n - 1];
                    itStart = stt + maxEDs;
                    itMode = 1;
                    if (stt == 0) itStart = 0;
                    else if (dirN == 1) itMode = 2;
                    else if (otherRefN != BFS_NONE) {
                        uint32_t oOff, oDesc, oInit;
                        if (otherSelf) {
                            oOff = aOff;
                            oDesc = P.nDescNew;
                            oInit = nInitNew;
                        } else {
                            const uint4 oC4 = B.C[(size_t)otherRefN * CTX_U4 + 4];
                            oOff = oC4.x;
                            oDesc = oC4.y & 0xFFu;
                            oInit = (oC4.y >> 8) & 0xFFu;
                        }
                        if (oDesc > 0) {
                            const uint16_t* oi = reinterpret_cast<const uint16_t*>(B.A + oOff + 2 * oDesc);
                            itStart -= oDesc - oInit + (uint32_t)oi[oInit - 1];
                        }
                    }
                    itMeta = packMeta(smShiftN, maxEDs, minEDs, stt == 0, ITEM_EDIT);
                }
                uint4* Cx = B.C + (size_t)cNew * CTX_U4;
                Cx[0] = make_uint4(rsId, g.n | (g.m << 16), g.Wv | (g.Wh << 8) | (maxEDn << 16) | (clSize << 24),
                                   idxN | (dirN << 4) | (uniN << 5) | (useRev << 6) | (itMode << 7) | (scheme << 12) |
                                       (search << 16));
                Cx[1] = make_uint4(itStart, itMeta, descRefN, otherRefN);
                if (rsId > 0x1FFFFFFu || itMeta > 0x7FFFFFu) flags |= FLAG_CAPACITY;
                Cx[CTX_HOT] = make_uint4(rsId | (itMode << 25) | (dirN << 27) | (uniN << 28),
                                   g.n | (g.m << 9) | (g.Wv << 18) | (g.Wh << 23) | (maxEDn << 27), itStart,
                                   itMeta | (clSize << 23));
                Cx[2] = make_uint4(smR.sa.b, smR.sa.e, smR.rev.b, smR.rev.e);
                Cx[3] = make_uint4(smDepthN, smShiftN, smDist, xOff | (xLen << 16));
                Cx[4] = make_uint4(descSelf || otherSelf ? aOff : 0u,
                                   descSelf || otherSelf ? (P.nDescNew | (nInitNew << 8)) : 0u, 0u, 0u);
                const uint32_t* Gs[4]; // the bit-strings of A, C, G, T for this read x strand and direction
                for (uint32_t c4 = 0; c4 < 4; c4++) Gs[c4] = gString(G, gw, rsId, (uint32_t)useRev, c4);
                for (uint32_t b = 0; b < nBlk; b++) {
                    const uint64_t a = matchWord(Gs[0], xOff, xLen, b), c = matchWord(Gs[1], xOff, xLen, b);
                    const uint64_t gg = matchWord(Gs[2], xOff, xLen, b), t = matchWord(Gs[3], xOff, xLen, b);
                    Cx[CTX_M + 2 * b] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)c, (uint32_t)(c >> 32));
                    Cx[CTX_M + 1 + 2 * b] = make_uint4((uint32_t)gg, (uint32_t)(gg >> 32), (uint32_t)t, (uint32_t)(t >> 32));
                }
                // first cell of the cluster (:452-461)
                EdPack pk{0, 0};
                uint32_t fcCur = BFS_NONE;
                if (g.inFinalColumn(0)) {
                    const uint32_t e0 = cellAt(0, xLen, HP, HN, score);
                    if (e0 > 31u) flags |= FLAG_CAPACITY;
                    edPut(pk, 0, min(e0, 31u));
                    uint4* Fr = B.F + (size_t)fNext * F_U4;
                    Fr[0] = make_uint4(smR.sa.b, smR.sa.e, smR.rev.b, smR.rev.e);
                    Fr[1] = make_uint4(0u, BFS_NONE, 0u, 0u);
                    fcCur = fNext++;
                }
                bool live = true;
                RangePair root = smR;
                uint32_t rootRow = 0;
                if (nSrcDesc > 0) { // replay (:463-492)
                    const uint32_t maxRow = g.m - 1;
                    for (uint32_t j = 0; j < nSrcDesc; j++) {
                        const uint32_t meta = dl[2 * j + 1].x;
                        const uint32_t depth = meta & 0xFFFFu, ch = (meta >> 16) & 0xFFu;
                        if (depth > maxRow) break;
                        const uint64_t M = matchWord(gString(G, gw, rsId, (uint32_t)useRev, ch - 1u), xOff, xLen, depth / MX_BLOCK);
                        uint64_t D0;
                        const bool valid = computeRow(g, depth, M, HP, HN, D0, RAC, score);
                        cRows++;
                        if (g.inFinalColumn(depth)) {
                            const uint32_t cellJ = clSize + depth - g.m;
                            const uint32_t e = cellAt(depth, g.n - 1, HP, HN, score);
                            if (e > 31u) flags |= FLAG_CAPACITY;
                            edPut(pk, cellJ, min(e, 31u));
                            uint4* Fr = B.F + (size_t)fNext * F_U4;
                            Fr[0] = dl[2 * j];
                            Fr[1] = make_uint4(depth | (ch << 16), fcCur, 0u, 0u);
                            fcCur = fNext++;
                            if (!valid || onlyVerticalGapsLeft(g, depth, HN)) { // goDeeper, then `return` (:472-477)
                                outKind = 2;
                                oEv = make_uint4(cNew, fcCur, j + 1, cellJ);
                                oEv1 = make_uint4((uint32_t)pk.lo, (uint32_t)(pk.lo >> 32), (uint32_t)pk.hi, (uint32_t)(pk.hi >> 32));
                                live = false;
                                break;
                            }
                        }
                        if (!valid) {
                            live = false;
                            break;
                        }
                    }
                    if (live) {
                        const uint32_t lastDepth = dl[2 * (nSrcDesc - 1) + 1].x & 0xFFFFu;
                        if (lastDepth == maxRow) live = false; // :479
                        else {
                            rootRow = lastDepth;
                            if (!dswN) { // after a switch the range of the start match is kept (:485)
                                const uint4 lr = dl[2 * (nSrcDesc - 1)];
                                root = RangePair{{lr.x, lr.y}, {lr.z, lr.w}};
                            }
                        }
                    }
                }
                if (live) {
                    outKind = 1;
                    oN0 = make_uint4(root.sa.b, root.sa.e, root.rev.b, root.rev.e);
                    oN1 = make_uint4(rootRow | (score << 16), cNew, fcCur, (uint32_t)__ffsll((unsigned long long)RAC) - 1u);
                    oN2 = make_uint4((uint32_t)HP, (uint32_t)(HP >> 32), (uint32_t)HN, (uint32_t)(HN >> 32));
                    oN4 = make_uint4((uint32_t)pk.lo, (uint32_t)(pk.lo >> 32), (uint32_t)pk.hi, (uint32_t)(pk.hi >> 32));
                }
            }
        }
        // ---- append the node / event this lane produced
        uint32_t t4, t5;
        const uint32_t oN = blockAppend(&B.nq[outP], outKind == 1 ? 1u : 0u, sh[0], t4);
        const uint32_t oE = blockAppend(&B.ne[outP], outKind == 2 ? 1u : 0u, sh[1], t5);
        if (outKind == 1) {
            if (oN >= qCap) flags |= FLAG_BFS_Q;
            else {
                Qo[oN] = oN0;
                Qo[(size_t)qCap + oN] = oN1;
                Qo[(size_t)2 * qCap + oN] = oN2;
                Qo[(size_t)3 * qCap + oN] = oN4;
            }
        } else if (outKind == 2) {
            if (oE >= B.evCap) flags |= FLAG_BFS_EV;
            else {
                Eo[(size_t)2 * oE] = oEv;
                Eo[(size_t)2 * oE + 1] = oEv1;
            }
        }
    }
    unsigned long long v = cRows;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((tid & 63u) == 0 && v) atomicAdd(&q.counters[11], v);
    if (flags) atomicOr(&q.cnt[3], flags);
}

// first approximate phase of every task (k_exact's DfsTask queue) -> frontier of pass 0
__global__ void __launch_bounds__(256)
k_bfs_start(DevIndex ix, const DevStrategyK* __restrict__ stp, BfsBufs B, const DfsTask* __restrict__ tasks, uint32_t nTasks,
            const uint64_t* __restrict__ offs, uint32_t gw, const uint32_t* __restrict__ G, const PartOut* __restrict__ parts,
            Queues q) {
    if (blockStopped(q)) return;
    bfsHeavy<true>(ix, stp, B, 0u, tasks, nTasks, offs, gw, G, parts, q, blockIdx.x, gridDim.x);
}

// one level: blocks [0, BFS_GRID) expand the frontier, blocks [BFS_GRID, BFS_GRID + BFS_GRID_EV) handle the events
// of the same pass (both only append to the queues of pass + 1, so they run side by side)
__global__ void __launch_bounds__(256)
k_bfs_pass(DevIndex ix, const DevStrategyK* __restrict__ stp, BfsBufs B, uint32_t pass, const uint64_t* __restrict__ offs,
           uint32_t gw, const uint32_t* __restrict__ G, const PartOut* __restrict__ parts, Queues q) {
    if (blockStopped(q)) return;
    if (blockIdx.x < BFS_GRID) bfsExpand(ix, B, pass, q, blockIdx.x, BFS_GRID);
    else bfsHeavy<false>(ix, stp, B, pass, nullptr, 0u, offs, gw, G, parts, q, blockIdx.x - BFS_GRID, BFS_GRID_EV);
}

} // namespace cmb

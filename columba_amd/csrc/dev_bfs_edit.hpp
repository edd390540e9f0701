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
#include "dev_partition.hpp"
#include "dev_wave.hpp"

namespace cmb {

constexpr uint32_t BFS_NONE = 0xFFFFFFFFu;
constexpr uint32_t ED_CELLS = 24; // final-column cells per phase (5 bits each in a 128-bit pack); 3k+2 <= 24 for k <= 7
// A context is four 128-byte lines: line 0 the cold header (C0..C4), line 1 the hot word and the match words of
// row blocks 0..2 — what an expansion reads of its context is ONE line for the first 96 rows of a phase (with the
// hot word in line 0 the kernel took 72 instead of 66 ms) —, lines 2 and 3 the match words of row blocks 3..10
// (a part of a 256-character read can span all of it: 256 + 20 rows; with three lines a part beyond 223 rows — one
// of the two parts of a long read at k = 1 — stopped the whole batch with CMB_ERR_INTERNAL).
constexpr uint32_t CTX_U4 = 32;    // uint4 per context for reads of up to 320 characters (BfsBufs::ctxU4 is what the kernels use)
constexpr uint32_t CTX_HOT = 8;   // uint4 index of the hot word
constexpr uint32_t CTX_M = 10;    // uint4 index of the match words of row block 0 ({A,C}, {G,T} per block)
constexpr uint32_t CTX_MBLK = 11; // row blocks with cached match words (rows < 352) in a context of CTX_U4
// longer reads (up to MAX_READ): contexts of 48 uint4 with the match words of 16 row blocks (rows < 512)
constexpr uint32_t CTX_U4_LONG = 48, CTX_MBLK_LONG = 16;
// GeoX (11 ... 13 errors: 16-row blocks): 80 uint4 with the match words of 35 row blocks (rows < 560)
constexpr uint32_t CTX_U4_X = 80, CTX_MBLK_X = 35;
__host__ __device__ inline uint32_t ctxU4For(uint32_t maxLen) { return maxLen > 320u ? CTX_U4_LONG : CTX_U4; }
__host__ __device__ inline uint32_t ctxMblkFor(uint32_t maxLen) { return maxLen > 320u ? CTX_MBLK_LONG : CTX_MBLK; }
constexpr uint32_t F_U4 = 2;      // uint4 per F record: {ranges} {depth | c << 16, parent, reported, -}
// (all blocks of a pass should be resident together — 3 blocks of 256 threads per CU at ~160 VGPRs — or the event
// blocks, which come last in the grid, only start when expansion blocks have finished)
constexpr uint32_t BFS_CHAIN = 4;     // expansions a lane makes in a row while each yields exactly one plain node
constexpr uint32_t BFS_GRID = 1024;   // MOST blocks that expand the frontier (grid-stride); sizes the per-block counters
constexpr uint32_t BFS_GRID_X = 896;  // default: expanding blocks (k_bfs_pass runs four 256-thread blocks per CU) ...
constexpr uint32_t BFS_GRID_EV = 128; // ... and blocks that handle the events of the same pass
constexpr uint32_t BFS_GRID_X_WALK = 384; // bfsExpandWalk (64 KB of LDS per block: two blocks per CU, all of a pass resident together)

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
    uint4* Q[2];  // frontier nodes, 4 planes of qCap: {ranges} {row | score << 16, ctx, fc, RAC bit | mode << 8} {HP, HN}
                  // {final-column distances of the path: only touched for nodes in the final column}
    uint4* Ev[2]; // events, 2 x 16 B: {ctx, F index of the node that ended its path, remaining-descendants index | -1,
                  // cell} {final-column distances of the path}
    uint4* F;     // final-column records, 2 x 16 B: {ranges} {depth | c << 16, parent, reported, -}
    uint4* C;     // contexts, ctxU4 x 16 B
    uint4* A;     // list arena: descendants (2 x 16 B each: ranges, {depth | c << 16}) and initial distances (u16)
    uint32_t qCap, evCap, fCap, cCap, aCap;
    uint32_t ctxU4, ctxMblk; // size of a context in uint4 / row blocks whose match words it caches (ctxU4For, ctxMblkFor)
    uint32_t chain;  // expansions a lane makes in a row while each yields exactly one plain node (BFS_CHAIN)
    uint32_t gridX, gridEv; // blocks of k_bfs_pass that expand / that handle events
    uint32_t* nq;    // [pass] number of frontier nodes consumed by pass `pass`
    uint32_t* ne;    // [pass] number of events consumed by pass `pass`
    uint32_t* pool;  // [0] F records, [1] contexts, [2] arena units handed out
    unsigned long long* blockCnt; // [BFS_GRID][4] per-block counters: nodes, expansions, rows, -
    // bfsExpandWalk: the node queues are made of chunks of 64 slots; qCnt[j][c] = nodes in chunk c of Q[j]
    uint32_t* qCnt[2];
    uint32_t* wcSave; // [wavefront][WALK_SAVE_U32]: the item and F chunks a wavefront carries from pass to pass
};

#ifdef CMB_BOUNDS
// diagnostic build only (tools/bounds_check.sh): the data-dependent indices of the frontier kernels are checked; the
// first violation is recorded ([0] site, [1] index, [2] capacity) and the index replaced by 0 instead of faulting
__device__ unsigned long long g_oob[4];
__device__ __forceinline__ uint32_t boundsChecked(uint32_t idx, uint32_t cap, uint32_t site) {
    if (idx >= cap) {
        if (atomicCAS(&g_oob[0], 0ull, (unsigned long long)site) == 0ull) {
            g_oob[1] = idx;
            g_oob[2] = cap;
        }
        return 0u;
    }
    return idx;
}
#define CMB_IDX(idx, cap, site) boundsChecked((idx), (cap), (site))
#else
// production: an index read from a record that lies outside its pool stops the search with CMB_ERR_INTERNAL (the
// enclosing function's `flags`) instead of faulting
#define CMB_IDX(idx, cap, site) ((idx) < (cap) ? (idx) : (flags |= FLAG_CAPACITY, 0u))
#endif

#ifdef CMB_BFS_STATS
// diagnostic build only (tools/bfs_stats.sh): what an expansion produces — [0] nothing, [1] exactly one node outside the
// final column and nothing else, [2] exactly one node in the final column and nothing else, [3] anything else,
// [4] of [1]: the child's row stays in the parent's 32-row matrix block, [5] expansions whose two ends share a rank block
__device__ unsigned long long g_bfsStats[16];
#endif

// What bfsHeavy (events, phase entry) needs to know about the index behind the search: the range-pair type of a node and
// how it is laid out in the records.  FmTraits: the FM-index (four 32-bit bounds in one uint4); the run-length compressed
// backend brings its own (move_search.hpp: ranges with run indices and a toehold, five uint4).  A pair occupies PAIR_U4
// uint4 at `stride` apart (1: F records, descendant lists; qCap: the planes of the node queue).
#ifndef CMB_BFS_WALK
#define CMB_BFS_WALK 0
#endif
struct FmTraits {
    typedef RangePair Pair;
    typedef DfsTask Task;
    static constexpr uint32_t PAIR_U4 = 1;
    static constexpr bool CHUNKED_Q = CMB_BFS_WALK != 0; // nodes go to chunks of 64 slots with a count each (bfsExpandWalk reads them)
    static __device__ __forceinline__ Pair load(const uint4* p, size_t) {
        const uint4 v = p[0];
        return RangePair{{v.x, v.y}, {v.z, v.w}};
    }
    static __device__ __forceinline__ void store(uint4* p, size_t, const Pair& r) { p[0] = make_uint4(r.sa.b, r.sa.e, r.rev.b, r.rev.e); }
    static __device__ __forceinline__ Pair none() { return RangePair{{0, 0}, {0, 0}}; }
    static __device__ __forceinline__ bool empty(const Pair& r) { return r.empty(); }
    static __device__ __forceinline__ Pair taskRange(const Task& t) { return t.r; }
    // an in-index occurrence (FMOcc) of read x strand rsId
    template <class BUFS>
    static __device__ __forceinline__ void emitFm(const BUFS&, const Queues& q, uint32_t slot, uint32_t rsId, const Pair& r, uint32_t depth,
                                                  uint32_t ed, uint32_t shift) {
        q.fm[slot] = FMOccRec{rsId, r.sa.b, r.sa.e, depth, ed, shift};
    }
    template <class BUFS> static __device__ __forceinline__ void fmHole(const BUFS&, const Queues& q, uint32_t slot) { q.fm[slot].rsId = 0xFFFFFFFFu; }
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

// The wide pack (edit distance 8 ... 10): a phase has up to 3 k + 2 = 32 final-column cells, and the distances below a cluster centre
// reach maxED + 31, so a cell takes 6 bits: ten cells per 64-bit word, four words.
struct EdPackW {
    uint64_t w[4];
};
__device__ __forceinline__ uint32_t edGet(const EdPackW& p, uint32_t i) {
    const uint32_t q = i / 10u, r = i - 10u * q;
    const uint64_t x = q == 0 ? p.w[0] : q == 1 ? p.w[1] : q == 2 ? p.w[2] : p.w[3];
    return (uint32_t)(x >> (6u * r)) & 63u;
}
__device__ __forceinline__ void edPut(EdPackW& p, uint32_t i, uint32_t v) { // cell i is still zero
    const uint32_t q = i / 10u, r = i - 10u * q;
    const uint64_t x = (uint64_t)v << (6u * r);
    if (q == 0) p.w[0] |= x;
    else if (q == 1) p.w[1] |= x;
    else if (q == 2) p.w[2] |= x;
    else p.w[3] |= x;
}
// a pack in the records: PK_U4 consecutive planes (nodes) or uint4 (events)
__device__ __forceinline__ void packLoad(const uint4* p, size_t stride, EdPack& out) {
    const uint4 a = p[0];
    out = EdPack{(uint64_t)a.x | ((uint64_t)a.y << 32), (uint64_t)a.z | ((uint64_t)a.w << 32)};
}
__device__ __forceinline__ void packStore(uint4* p, size_t stride, const EdPack& k) {
    p[0] = make_uint4((uint32_t)k.lo, (uint32_t)(k.lo >> 32), (uint32_t)k.hi, (uint32_t)(k.hi >> 32));
}
__device__ __forceinline__ void packLoad(const uint4* p, size_t stride, EdPackW& out) {
    const uint4 a = p[0], b = p[stride];
    out.w[0] = u64of(a.x, a.y), out.w[1] = u64of(a.z, a.w), out.w[2] = u64of(b.x, b.y), out.w[3] = u64of(b.z, b.w);
}
__device__ __forceinline__ void packStore(uint4* p, size_t stride, const EdPackW& k) {
    p[0] = make_uint4((uint32_t)k.w[0], (uint32_t)(k.w[0] >> 32), (uint32_t)k.w[1], (uint32_t)(k.w[1] >> 32));
    p[stride] = make_uint4((uint32_t)k.w[2], (uint32_t)(k.w[2] >> 32), (uint32_t)k.w[3], (uint32_t)(k.w[3] >> 32));
}
// The two geometries of the frontier's records: GeoN is the common path (tables of MAXP parts, 24 cells of 5 bits: k <= 7) — every
// kernel of the headline path is the GeoN instance, with the registers and record sizes it always had; GeoW: tables of MAXP_WIDE parts,
// 32 cells of 6 bits (k = 8 ... 10), one more plane per node and one more uint4 per event.
// The in-index matrix of a geometry: the reference's 64-bit matrix (32-row blocks: up to 10 errors), or — GeoX, 11 ... 13 errors — the
// 64-bit matrix with 16-row blocks that stands in for the reference's 64- AND 128-bit matrices (dev_matrix.hpp: MXN_*).
struct MxRef64 {
    static constexpr uint32_t BLOCK = MX_BLOCK, LEFT = MX_LEFT, DIAG = MX_DIAG;
    static __device__ __forceinline__ bool row(const MatGeom& g, uint32_t i, uint64_t M, uint64_t& HP, uint64_t& HN, uint64_t& D0, uint64_t& RAC,
                                               uint32_t& sc) {
        return computeRow(g, i, M, HP, HN, D0, RAC, sc);
    }
    static __device__ __forceinline__ uint32_t cell(uint32_t i, uint32_t j, uint64_t HP, uint64_t HN, uint32_t sc) { return cellAt(i, j, HP, HN, sc); }
    static __device__ __forceinline__ bool ovgl(const MatGeom& g, uint32_t i, uint64_t HN) { return onlyVerticalGapsLeft(g, i, HN); }
};
struct MxNarrow {
    static constexpr uint32_t BLOCK = MXN_BLOCK, LEFT = MXN_LEFT, DIAG = MXN_DIAG;
    static __device__ __forceinline__ bool row(const MatGeom& g, uint32_t i, uint64_t M, uint64_t& HP, uint64_t& HN, uint64_t& D0, uint64_t& RAC,
                                               uint32_t& sc) {
        return computeRowWide<BLOCK, DIAG>(g, i, M, HP, HN, D0, RAC, sc); // (the walk to the rightmost active column spans up to 39 columns)
    }
    static __device__ __forceinline__ uint32_t cell(uint32_t i, uint32_t j, uint64_t HP, uint64_t HN, uint32_t sc) {
        return cellAt<BLOCK, DIAG>(i, j, HP, HN, sc);
    }
    static __device__ __forceinline__ bool ovgl(const MatGeom& g, uint32_t i, uint64_t HN) { // (as the part's own matrix would answer)
        return onlyVerticalGapsLeftAs<BLOCK, DIAG>(g, i, HN, g.maxED <= MX_MAX_ED ? 64u : 128u);
    }
};
struct GeoN : MxRef64 {
    static constexpr int MP = MAXP;
    typedef EdPack Pack;
    static constexpr uint32_t CELLS = 24, PK_U4 = 1, ED_MAX = 31;
};
struct GeoW : MxRef64 {
    static constexpr int MP = MAXP_WIDE;
    typedef EdPackW Pack;
    static constexpr uint32_t CELLS = 32, PK_U4 = 2, ED_MAX = 63;
};
struct GeoX : MxNarrow { // 11 ... 13 errors: final columns of up to 3 * 13 + 1 = 40 cells (values up to 13 + 39)
    static constexpr int MP = MAXP_WIDE;
    typedef EdPackW Pack;
    static constexpr uint32_t CELLS = 40, PK_U4 = 2, ED_MAX = 63;
};

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

// the same into a queue made of chunks of 64 slots with a count each (the node queue that bfsExpandWalk reads): the block takes whole
// chunks and writes their counts
__device__ __forceinline__ uint32_t blockAppendChunked(uint32_t* counter, uint32_t n, uint32_t* sh, uint32_t& blockTotal, uint32_t* qCnt,
                                                       uint32_t cap) {
    uint32_t wTotal;
    const uint32_t pre = waveExclusiveScan(n, wTotal);
    const uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) sh[w] = wTotal;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = sh[0] + sh[1] + sh[2] + sh[3];
        sh[4] = t ? atomicAdd(counter, (t + 63u) & ~63u) : 0u;
    }
    __syncthreads();
    uint32_t off = sh[4];
    blockTotal = sh[0] + sh[1] + sh[2] + sh[3];
    if (threadIdx.x < 4 && threadIdx.x * 64u < blockTotal && off + threadIdx.x * 64u < cap)
        qCnt[(off >> 6) + threadIdx.x] = min(64u, blockTotal - threadIdx.x * 64u);
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

// the two rank blocks an extension of `p` in `mode` needs: request (raw 16-byte chunks) and use
__device__ __forceinline__ void issueRanks(const DevIndex& ix, int mode, const RangePair& p, uint4 v[4]) {
    DevBWT t = ix.fwd; // values, not references, are selected
    Range tr = p.sa;
    if (mode == 0) {
        t = ix.rev;
        tr = p.rev;
    }
#ifdef CMB_BOUNDS
    tr.b = CMB_IDX(tr.b, ix.n + 2u, 200);
    tr.e = CMB_IDX(tr.e, ix.n + 2u, 201);
#endif
    loadRankPairRaw(t, tr.b, tr.e, v);
}
__device__ __forceinline__ void takeRanks(const DevIndex& ix, int mode, const RangePair& p, const uint4 v[4], uint32_t Rb[4],
                                          uint32_t Re[4], uint32_t& db, uint32_t& de) {
    const uint32_t dollar = mode == 0 ? ix.rev.dollarPos : ix.fwd.dollarPos;
    const uint32_t b = mode == 0 ? p.rev.b : p.sa.b, e = mode == 0 ? p.rev.e : p.sa.e;
    uint4 w[2];
    rankPairEnd(v, b, e, w);
    ranksFromRaw(v, b, dollar, Rb);
    ranksFromRaw(w, e, dollar, Re);
    db = b > dollar ? 1u : 0u;
    de = e > dollar ? 1u : 0u;
}

// ------------------------------------------------------------------ expand
// One lane per frontier node.  Children (extendFMPos, indexinterface.cpp:675-697) get their matrix row at
// once (computeRow; the reference computes it when the child is popped — every pushed child is popped) and
// are classified as in branchAndBound (:529-561) + the stack loop (:506-526):
//   row invalid outside the final column      -> dropped
//   final column, row invalid or only vertical gaps left -> event (goDeeper)
//   narrow range (in-text switch, :516)        -> in-text verification items
//   otherwise                                  -> node of the next frontier
//
// The kernel waits for scattered memory, so what it can keep in flight is set by its registers (measured: 2 / 3
// wavefronts per SIMD = 99 / 71 ms per step).  Hence a lane never HOLDS the four children: it classifies them
// (evalChild<false>: 4 bits per child), the block allocates the queue slots, and the children that leave are
// computed again (evalChild<true>, ~50 VALU instructions each) straight into their records.
enum : uint32_t { KIND_NONE = 0, KIND_NODE = 1, KIND_EVENT = 2, KIND_ITEMS = 3 };
struct ExpandCtx { // what the expansions of one phase share (from the context's hot word)
    MatGeom g;
    uint32_t clSize, itMode, itStart, switchPoint;
};
struct ChildState {
    RangePair r;
    uint64_t HP, HN, RAC;
    uint32_t sc, aux; // aux: final-column distance of the child (needF) or in-text start difference (KIND_ITEMS)
};
// kind of a (non-empty) child at row `row1` | needF << 2 | capacity problem << 3, and its state: the child's matrix row (computeRow),
// branchAndBound (:529-561), the in-text switch (:340-375, :516)
template <class Geo = GeoN>
__device__ __forceinline__ uint32_t evalRow(const RangePair& child, const ExpandCtx& e, uint32_t row1, bool inFC, uint64_t M, uint64_t pHP,
                                            uint64_t pHN, uint64_t pRAC, uint32_t score, ChildState& out) {
    uint64_t HP = pHP, HN = pHN, RAC = pRAC, D0;
    uint32_t sc = score;
    constexpr uint32_t EDMAX = Geo::ED_MAX;
    const bool valid = Geo::row(e.g, row1, M, HP, HN, D0, RAC, sc);
    if (!valid && !inFC) return KIND_NONE; // pruned when popped (branchAndBound returns true, :560)
    uint32_t res = KIND_NODE, aux = 0;
    if (inFC) {
        const uint32_t ed = Geo::cell(row1, e.g.n - 1, HP, HN, sc);
        aux = min(ed, EDMAX);
        res |= 4u;
        if (ed > EDMAX) res |= 8u;
        if (!valid || Geo::ovgl(e.g, row1, HN)) res = (res & ~3u) | KIND_EVENT;
    }
    if ((res & 3u) == KIND_NODE && child.sa.width() <= e.switchPoint && e.itMode != 0) { // goToInTextVerificationEdit (:340-375)
        uint32_t startDiff = e.itStart;
        if (e.itMode == 2) {
            const uint32_t col = e.g.firstColumn(row1);
            startDiff -= col + Geo::cell(row1, col, HP, HN, sc);
        }
        aux = startDiff;
        res = (res & 8u) | KIND_ITEMS; // (a child that leaves the index needs no F record)
    }
    out.r = child;
    out.HP = HP;
    out.HN = HN;
    out.RAC = RAC;
    out.sc = sc;
    out.aux = aux;
    return res;
}
// kind of child `ch` of `parent` at row `row1` | needF << 2 | capacity problem << 3; FULL: also its state
template <bool FULL, class Geo = GeoN>
__device__ __forceinline__ uint32_t evalChild(const DevIndex& ix, int md, const RangePair& parent, uint32_t ch,
                                              const uint32_t Rb[4], const uint32_t Re[4], uint32_t db, uint32_t de,
                                              const ExpandCtx& e, uint32_t row1, bool inFC, uint64_t M, uint64_t pHP,
                                              uint64_t pHN, uint64_t pRAC, uint32_t score, bool& nonEmpty, ChildState& out) {
    RangePair child;
    nonEmpty = childFromRanks(ix, md, parent, ch, Rb, Re, db, de, child);
    if (!nonEmpty) return KIND_NONE;
    ChildState cs;
    const uint32_t res = evalRow<Geo>(child, e, row1, inFC, M, pHP, pHN, pRAC, score, cs);
    if (FULL && (res & 3u) != KIND_NONE) out = cs;
    return res;
}

template <class Geo = GeoN>
__device__ __forceinline__ void bfsExpand(const DevIndex& ix, const BfsBufs& B, uint32_t pass, const Queues& q,
                                          uint32_t bid, uint32_t nBlocks) {
    __shared__ uint32_t sh[4][5];
    // per-lane state that is touched once or twice per expansion lives in LDS, [field][lane], not in registers: the
    // four match words of the row block, the eight ranks of the pending expansion, the node's final-column pack
    __shared__ uint64_t ldsM[4][256];
    __shared__ uint32_t ldsR[8][256];
    __shared__ uint32_t ldsCnt[2][256]; // per-lane counters: children (= matrix rows), expansions
    const uint32_t tid = threadIdx.x;
    ldsCnt[0][tid] = 0;
    ldsCnt[1][tid] = 0;
    const uint32_t nIn = min(B.nq[pass], B.qCap);
    const uint4* __restrict__ Qi = B.Q[pass & 1u];
    uint4* __restrict__ Qo = B.Q[(pass + 1u) & 1u];
    uint4* __restrict__ Eo = B.Ev[(pass + 1u) & 1u];
    const uint32_t qCap = B.qCap;
    uint32_t flags = 0;
    for (uint32_t base = bid * 256u; base < nIn; base += nBlocks * 256u) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        const bool act = i < nIn;
        uint32_t kinds = 0; // 4 bits per child: kind | needF << 2
        uint32_t row1 = 0, ctx = 0, fcP = BFS_NONE, nIt = 0;
        // ---- the node and what the expansions of its phase need of its context: one memory step
        RangePair parent{{0, 0}, {0, 0}};
        uint32_t row = 0, score = 0, blk = 0;
        int md = 0;
        uint64_t pHP = 0, pHN = 0;
        uint32_t pRac = 0; // (RAC is always a single bit: kept as its index)
        ExpandCtx e{};
        e.switchPoint = ix.switchPoint;
        uint32_t db = 0, de = 0;
        bool walking = act;
        if (act) {
            const uint4 n0 = Qi[i], n1 = Qi[(size_t)qCap + i], n2 = Qi[(size_t)2 * qCap + i];
            ctx = n1.y;
            fcP = n1.z;
            row = n1.x & 0xFFFFu;
            score = n1.x >> 16;
            // the node carries the extension mode of its phase, so that the rank blocks of its first expansion are
            // requested together with the context words (one round trip less than fetching the mode from the context)
            md = (int)((n1.w >> 8) & 3u);
            parent = RangePair{{n0.x, n0.y}, {n0.z, n0.w}};
            uint4 rk[4];
            issueRanks(ix, md, parent, rk);
            const uint4* Cx = B.C + (size_t)CMB_IDX(ctx, B.cCap, 1) * B.ctxU4;
            blk = (row + 1) / Geo::BLOCK;
            const uint4 hot = Cx[CTX_HOT]; // everything the expansion needs of its context, in ONE 16-byte request
            const uint4 mA = Cx[CTX_M + 2 * blk], mB = Cx[CTX_M + 1 + 2 * blk];
            {
                uint32_t Rb[4], Re[4];
                takeRanks(ix, md, parent, rk, Rb, Re, db, de);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    ldsR[c][tid] = Rb[c];
                    ldsR[4 + c][tid] = Re[c];
                }
            }
            e.itStart = hot.z;
            e.g.n = hot.y & 0x1FFu;
            e.g.m = (hot.y >> 9) & 0x1FFu;
            e.g.Wv = (hot.y >> 18) & 31u;
            e.g.Wh = (hot.y >> 23) & 15u;
            e.g.maxED = (hot.y >> 27) & 15u;
            e.clSize = hot.w >> 23;
            e.itMode = (hot.x >> 25) & 3u; // 0: phase 0 (no switch), 1: start difference fixed, 2: BACKWARD
            pHP = u64of(n2.x, n2.y);
            pHN = u64of(n2.z, n2.w);
            pRac = n1.w & 63u;
            ldsM[0][tid] = u64of(mA.x, mA.y);
            ldsM[1][tid] = u64of(mA.z, mA.w);
            ldsM[2][tid] = u64of(mB.x, mB.y);
            ldsM[3][tid] = u64of(mB.z, mB.w);
        }
        // ---- walk: three expansions in four produce exactly one plain node and nothing else (measured, DESIGN.md
        // §4.2).  Such a child is expanded at once by the same lane — no node record written and read back, no queue
        // slot, no context fetch — for up to B.chain rows inside one 32-row matrix block; the lane stops at the
        // first expansion that produces anything else and keeps its PARENT state: the children are computed again
        // after the queue slots have been handed out.
        for (uint32_t step = 0; step < B.chain; step++) { // (wave-uniform exit below)
            if (walking) {
                if (step) {
                    uint4 rk[4];
                    issueRanks(ix, md, parent, rk); // the memory step of an expansion: two rank blocks
                    uint32_t Rb[4], Re[4];
                    takeRanks(ix, md, parent, rk, Rb, Re, db, de);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        ldsR[c][tid] = Rb[c];
                        ldsR[4 + c][tid] = Re[c];
                    }
                }
                uint32_t Rb[4], Re[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    Rb[c] = ldsR[c][tid];
                    Re[c] = ldsR[4 + c][tid];
                }
                row1 = row + 1;
                ldsCnt[1][tid] += 1u;
                const bool inFC = e.g.inFinalColumn(row1);
                if (inFC && e.clSize + row1 - e.g.m >= Geo::CELLS) flags |= FLAG_CAPACITY; // (a row beyond the matrix)
                kinds = 0;
                nIt = 0;
                uint32_t nOut = 0, nChildren = 0;
#pragma unroll
                for (uint32_t ch = 1; ch <= 4; ch++) {
                    bool nonEmpty;
                    ChildState cs;
                    const uint32_t k4 = evalChild<false, Geo>(ix, md, parent, ch, Rb, Re, db, de, e, row1, inFC, ldsM[ch - 1][tid], pHP,
                                                         pHN, 1ull << pRac, score, nonEmpty, cs);
                    nChildren += nonEmpty ? 1u : 0u;
                    if (k4 & 8u) flags |= FLAG_CAPACITY;
                    if ((k4 & 3u) == KIND_NONE) continue;
                    kinds |= (k4 & 7u) << (4 * (ch - 1));
                    nOut++;
                    if ((k4 & 3u) == KIND_ITEMS) { // (its width: the child's range again, a few instructions)
                        RangePair child;
                        (void)childFromRanks(ix, md, parent, ch, Rb, Re, db, de, child);
                        nIt += child.sa.e - child.sa.b;
                    }
                }
                ldsCnt[0][tid] += nChildren;
                // exactly one child, a plain node, with rows left in this matrix block: keep walking
                const bool single = nOut == 1u && (kinds == 0x1u || kinds == 0x10u || kinds == 0x100u || kinds == 0x1000u);
                if (single && step + 1u < B.chain && (row1 + 1u) / Geo::BLOCK == blk) {
                    const uint32_t ch = ((31u - (uint32_t)__clz(kinds)) >> 2) + 1u;
                    const uint64_t M = ch == 1 ? ldsM[0][tid] : ch == 2 ? ldsM[1][tid] : ch == 3 ? ldsM[2][tid] : ldsM[3][tid];
                    bool nonEmpty;
                    ChildState one;
                    (void)evalChild<true, Geo>(ix, md, parent, ch, Rb, Re, db, de, e, row1, inFC, M, pHP, pHN, 1ull << pRac, score,
                                          nonEmpty, one);
                    parent = one.r;
                    score = one.sc;
                    pHP = one.HP;
                    pHN = one.HN;
                    pRac = (uint32_t)__ffsll((unsigned long long)one.RAC) - 1u;
                    row = row1;
                    kinds = 0; // (nothing of this expansion is left to append)
                } else {
                    walking = false;
                }
            }
            if (__ballot(walking) == 0ull) break;
        }
        // ---- block-wide allocation in the four output queues
        uint32_t nNode = 0, nEv = 0, nF = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t kd = (kinds >> (4 * c)) & 3u;
            nNode += kd == KIND_NODE;
            nEv += kd == KIND_EVENT;
            nF += (kinds >> (4 * c + 2)) & 1u;
        }
#ifdef CMB_BFS_STATS
        if (act) {
            const uint32_t outs = nNode + nEv + (nIt ? 1u : 0u);
            const int cat = outs == 0 ? 0 : (outs == 1 && nNode == 1 && nF == 0) ? 1 : (outs == 1 && nNode == 1) ? 2 : 3;
            atomicAdd(&g_bfsStats[cat], 1ull);
        }
#endif
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
        if (kinds != 0u && okF && okNode && okEv && okIt) {
            const bool inFC = e.g.inFinalColumn(row1);
            const uint32_t cell = min(e.clSize + row1 - e.g.m, Geo::CELLS - 1u);
            // what only the records need is fetched again now: the final-column distances of the path so far (nodes in
            // the final column: one in thirty), read number and item word of the context (in-text items)
            typename Geo::Pack pack{};
            if (fcP != BFS_NONE && (kinds & 0x4444u)) packLoad(Qi + (size_t)3 * qCap + i, qCap, pack);
            uint32_t rsId = 0, itMeta = 0;
            if (nIt) {
                const uint4 hot = B.C[(size_t)CMB_IDX(ctx, B.cCap, 2) * B.ctxU4 + CTX_HOT];
                rsId = hot.x & 0x1FFFFFFu;
                itMeta = hot.w & 0x7FFFFFu;
            }
            uint32_t Rb[4], Re[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                Rb[c] = ldsR[c][tid];
                Re[c] = ldsR[4 + c][tid];
            }
#pragma unroll
            for (uint32_t ch = 1; ch <= 4; ch++) {
                const uint32_t kd = (kinds >> (4 * (ch - 1))) & 3u;
                if (kd == KIND_NONE) continue;
                bool nonEmpty;
                ChildState cs;
                (void)evalChild<true, Geo>(ix, md, parent, ch, Rb, Re, db, de, e, row1, inFC, ldsM[ch - 1][tid], pHP, pHN, 1ull << pRac,
                                      score, nonEmpty, cs);
                const bool wantF = (kinds >> (4 * (ch - 1) + 2)) & 1u;
                const uint4 cr = make_uint4(cs.r.sa.b, cs.r.sa.e, cs.r.rev.b, cs.r.rev.e);
                uint32_t fc = BFS_NONE;
                if (wantF) {
                    fc = oF++;
                    uint4* Fr = B.F + (size_t)CMB_IDX(fc, B.fCap, 9) * F_U4;
                    Fr[0] = cr;
                    Fr[1] = make_uint4(row1 | (ch << 16), fcP, 0u, 0u);
                }
                if (kd == KIND_NODE) {
                    const uint32_t o = oNode++;
                    Qo[o] = cr;
                    Qo[(size_t)qCap + o] = make_uint4(row1 | (cs.sc << 16), ctx, fc,
                                                      ((uint32_t)__ffsll((unsigned long long)cs.RAC) - 1u) | ((uint32_t)md << 8));
                    Qo[(size_t)2 * qCap + o] = make_uint4((uint32_t)cs.HP, (uint32_t)(cs.HP >> 32), (uint32_t)cs.HN,
                                                          (uint32_t)(cs.HN >> 32));
                    if (wantF) {
                        typename Geo::Pack p2 = pack;
                        edPut(p2, cell, cs.aux);
                        packStore(Qo + (size_t)3 * qCap + o, qCap, p2);
                    }
                } else if (kd == KIND_EVENT) {
                    typename Geo::Pack p2 = pack;
                    edPut(p2, cell, cs.aux);
                    constexpr uint32_t EV_U4 = 1u + Geo::PK_U4;
                    Eo[(size_t)EV_U4 * oEv] = make_uint4(ctx, fc, 0xFFFFFFFFu, cell);
                    packStore(Eo + (size_t)EV_U4 * oEv + 1, 1, p2);
                    oEv++;
                } else {
                    const uint32_t w = cs.r.sa.e - cs.r.sa.b;
                    for (uint32_t t = 0; t < w; t++) q.items[oIt + t] = make_uint4(rsId, cs.r.sa.b + t, cs.aux, itMeta);
                    oIt += w;
                }
            }
        }
    }
    // per-block counters (summed by k_bfs_finish): one writer per slot and launch, launches are ordered
    unsigned long long v[3] = {ldsCnt[0][tid], ldsCnt[1][tid], ldsCnt[0][tid]}; // (every child gets its matrix row)
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

// ------------------------------------------------------------------ expand: walking lanes over compacted children (round 4)
// What bounds bfsExpand above is not the memory system but INSTRUCTION ISSUE (round 4: one round of the kernel costs a wavefront
// ~20 000 cycles of which ~3 000 are the wait for its loads).  Two things waste issue slots there: of the four children every lane
// evaluates — child range, matrix row, classification, ~150 instructions each — only 1.7 exist on average (the other ranges are empty, but
// a wavefront executes the code while ANY lane needs it), and the chain steps run with ever fewer lanes.  Here a wavefront works in rounds
// of ONE memory round trip, on its own (no block barrier, no same-address atomic in the loop):
//   parents   every lane holds a node.  Its two rank blocks arrive, the lane computes the four child RANGES (cheap) and puts the
//             non-empty ones on the wavefront's CHILD LIST in LDS (prefix sum over the lanes: compaction);
//   children  the list is worked off 64 children at a time, one child per lane: the parent's row state, geometry and match word come
//             from the parent's LDS slots, the lane computes the child's matrix row and classifies it (evalRow — exactly what
//             bfsExpand does per child).  1.7 instead of 4 evaluations per expansion, all lanes busy;
//   walking   the first plain node among the children of a parent CLAIMS the parent's lane (an LDS atomic): it is expanded by that lane
//             in the next round — no node record, no queue slot, no context fetch, also across the 32-row matrix blocks (the match
//             words of the new block arrive with the rank blocks).  Its siblings and every other kind of child (final column, event,
//             in-text items) leave through the queues at once, slots by one prefix sum over the child lanes;
//   refill    the node a lane takes up when its walk ends WAITS IN LDS: global_load_lds copies the three node planes into the lane's
//             own slot ([plane][lane], no vector register) a round ahead, so that every lane expands in every round;
//   queues    input: the node queue is made of CHUNKS of 64 slots with a count each (BfsBufs::qCnt), chunk c belongs to wavefront
//             c % W — no atomics, no holes to skip.  Output: a wavefront fills a chunk of its own (one atomic per 64 nodes); events
//             go to small per-wavefront chunks whose unused rest becomes holes (context index 0xFFFFFFFF); in-text items and F
//             records to per-wavefront chunks that are kept ACROSS the passes (BfsBufs::wcSave; k_bfs_finish turns what is left of
//             the item chunks into holes);
//   tail      when a wavefront's input is used up its lanes walk on for a bounded number of rounds only (the emptier the machine, the
//             longer: a pass of a few thousand nodes finishes whole phases), then write their node out.
// Per child the semantics are those of bfsExpand line by line (same evalRow, same records, same counters); which lane expands a node,
// and in which pass, is not part of the result.
#ifndef CMB_BFS_WALK
#define CMB_BFS_WALK 0
#endif
constexpr uint32_t WALK_CH_EV = 32, WALK_CH_EV_SMALL = 4, WALK_CH_IT = 512, WALK_CH_F = 512;
constexpr uint32_t WALK_TAIL_MAX = 64;
constexpr uint32_t WALK_SAVE_U32 = 8; // per wavefront: the item chunk and the F chunk {base, used, size}, two spare words

// 16 bytes per active lane, global -> ldsWaveBase[lane], no vector register in between (global_load_lds_dwordx4).  Inline assembly, not
// __builtin_amdgcn_global_load_lds: beside the builtin hipcc 7.2 waits for ALL outstanding vector memory operations — the stores of the
// previous round included — before every LDS read that might alias and before the next such load (three extra round trips per round,
// measured).  The kernel waits for these loads itself: ONE s_waitcnt vmcnt(0) per round.
__device__ __forceinline__ void gldsU4(const uint4* src, uint4* ldsWaveBase) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_ptr_t)ldsWaveBase);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst)
                 : "memory");
}
#define CMB_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory") /* lanes of ONE wavefront talking through LDS */

// The wavefront's output REGION of the chunked node queue: a run of whole chunks (64 slots each) taken with ONE atomic and filled from
// the front; when it is left — used up, or at the end of the kernel — every chunk of it gets its count in qCnt (64, the rest, or 0: the
// consumers pass over empty chunks).  One chunk per atomic was too little: 1.8 million atomics per sub-batch on the queue's counter
// run into the ~90 same-address atomics per microsecond the L2 serves (the frontier kernel spent a third of its time queueing there).
struct NodeRegion {
    uint32_t base = 0xFFFFFFFFu, used = 0u, size = 0u;
    __device__ __forceinline__ void retire(uint32_t* qCnt) {
        const uint32_t lane = threadIdx.x & 63u;
        if (base != 0xFFFFFFFFu)
            for (uint32_t j = lane; j * 64u < size; j += 64u) qCnt[(base >> 6) + j] = used > j * 64u ? min(used - j * 64u, 64u) : 0u;
        base = 0xFFFFFFFFu;
        used = size = 0u;
    }
    // first slot for the wavefront's `total` nodes (contiguous); 0xFFFFFFFF: the queue is full.  `slots`: size of a new region (x 64)
    __device__ __forceinline__ uint32_t alloc(uint32_t* counter, uint32_t cap, uint32_t total, uint32_t slots, uint32_t* qCnt, bool& overflow) {
        if (used + total > size) {
            retire(qCnt);
            const uint32_t want = max(slots, (total + 63u) & ~63u);
            uint32_t b = 0;
            if ((threadIdx.x & 63u) == 0) b = atomicAdd(counter, want);
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            if (b > cap || want > cap - b) {
                overflow = true;
                return 0xFFFFFFFFu;
            }
            base = b;
            size = want;
            used = 0;
        }
        const uint32_t o = base + used;
        used += total;
        return o;
    }
};
__device__ __forceinline__ uint32_t nodeRegionSlots(uint32_t nIn, uint32_t W) { // what a wavefront is likely to append in a pass, 64 ... 512
    return min(512u, max(64u, ((nIn / max(W, 1u)) + 63u) & ~63u));
}

template <class Geo = GeoN>
__device__ __forceinline__ void bfsExpandWalk(const DevIndex& ix, const BfsBufs& B, uint32_t pass, const Queues& q,
                                              uint32_t bid, uint32_t nBlocks) {
    constexpr uint32_t PK = Geo::PK_U4, EV_U4 = 1u + Geo::PK_U4;
    __shared__ uint4 ldsNext[3][256];  // the node that waits for its lane: {ranges} {meta} {HP, HN}
    __shared__ uint4 ldsM4[2][256];    // match words of the lane's row block: {A, C} {G, T}
    __shared__ uint4 ldsPk[PK][256];   // final-column pack of a fresh node that already lies in the final column
    __shared__ uint4 pA[2][256];       // row state {HP, HN} of the lane's node, by round parity (children read, the heir writes)
    __shared__ uint2 pB[2][256];       // {row | score << 16, RAC bit | mode << 8}, likewise
    __shared__ uint4 pG[256];          // hot word of the node's context
    __shared__ uint4 pF[256];          // {context, F record of the nearest final-column node above, has a pack, -}
    __shared__ uint4 aR[256];          // ranges of the child the lane walks on with
    __shared__ uint32_t k4[256];       // bit 0: a child has claimed the lane
    __shared__ uint4 clR[4][256];      // child list of the wavefront: ranges ...
    __shared__ uint8_t clT[4][256];    // ... and parent lane | (character - 1) << 6
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wbase = tid & ~63u, wv = tid >> 6;
    const uint32_t nIn = min(B.nq[pass], B.qCap);
    const uint4* __restrict__ Qi = B.Q[pass & 1u];
    uint4* __restrict__ Qo = B.Q[(pass + 1u) & 1u];
    uint4* __restrict__ Eo = B.Ev[(pass + 1u) & 1u];
    const uint32_t* __restrict__ cntIn = B.qCnt[pass & 1u];
    uint32_t* __restrict__ cntOut = B.qCnt[(pass + 1u) & 1u];
    const uint32_t qCap = B.qCap;
    // input: chunk c of the node queue belongs to wavefront c % W
    const uint32_t W = nBlocks * 4u, wId = bid * 4u + wv, nChunks = (nIn + 63u) >> 6;
    // wave-uniform: entries [cUsed, cCnt) of `chunk` are left; nCnt, n2: the counts of chunk + W and of chunk + 2 W (n2 is fetched with
    // every round's memory step, so that a count is there before its chunk becomes the next one)
    uint32_t chunk = wId, cUsed = 0u, cCnt = 0u, nCnt = 0u, n2 = 0u;
    const uint32_t cntLast = nChunks ? nChunks - 1u : 0u;
    {
        const uint32_t c0 = cntIn[min(chunk, cntLast)], c1 = cntIn[min(chunk + W, cntLast)], c2 = cntIn[min(chunk + 2u * W, cntLast)];
        cCnt = chunk < nChunks ? (uint32_t)__builtin_amdgcn_readfirstlane((int)min(c0, 64u)) : 0u;
        nCnt = chunk + W < nChunks ? (uint32_t)__builtin_amdgcn_readfirstlane((int)min(c1, 64u)) : 0u;
        n2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(c2, 64u));
    }
    // rounds a lane may still walk once the wavefront's input is used up
    uint32_t tail = B.chain;
    if (nIn < W * 64u) tail = min(WALK_TAIL_MAX, max(B.chain, (W * 64u / max(nIn, 1u)) * B.chain));
    tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)tail);
    const uint32_t chEv = nIn < 262144u ? WALK_CH_EV_SMALL : WALK_CH_EV;
    // per-lane state of a parent
    bool have = false, nx = false, fresh = false, needM = false;
    RangePair parent{{0, 0}, {0, 0}};
    uint32_t row = 0, ctx = 0, fcP = BFS_NONE, nxIdx = 0, curIdx = 0, blk = 0, hotY = 0, clSize = 0;
    int md = 0;
    uint32_t cntChildren = 0, cntExp = 0, flags = 0;
    uint32_t cur = 0; // round parity (wave-uniform)
    NodeRegion ncNode;
    const uint32_t regionSlots = nodeRegionSlots(nIn, W);
    WaveChunk wcEv, wcIt, wcF;
    uint32_t* save = B.wcSave + (size_t)wId * WALK_SAVE_U32;
    { // the item and F chunks of the wavefront's previous pass (zeroed before the search)
        wcIt.base = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[0]);
        wcIt.used = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[1]);
        wcIt.size = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[2]);
        wcF.base = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[4]);
        wcF.used = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[5]);
        wcF.size = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[6]);
    }
#ifdef CMB_BFS_STATS
    // diagnostic build only (tools/walk_stats.sh): [0] rounds, [1] lanes that expand, [2] cycles in the memory wait, [3] cycles in the
    // loop, [4] turns of the child loop, [5] lanes that walk on, [6] nodes taken up, [7] children, [8..13] cycles: take + request | issue
    // of the memory step | parents | children | parents again | -
    unsigned long long pf[16] = {};
    const long long pfT0 = clock64();
    long long pfT = pfT0;
#define PF_LAP(j)                                        \
    {                                                    \
        const long long now_ = clock64();                \
        if ((tid & 63u) == 0) pf[j] += (unsigned long long)(now_ - pfT); \
        pfT = now_;                                      \
    }
#define PF_ADD(j, x) pf[j] += (x);
#else
#define PF_LAP(j)
#define PF_ADD(j, x)
#endif
    auto holeEv = [&](uint32_t o) { Eo[(size_t)EV_U4 * o] = make_uint4(BFS_NONE, 0u, 0u, 0u); };
    auto holeIt = [&](uint32_t o) { q.items[o] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u); };
    auto holeF = [&](uint32_t) {}; // (the pool of F records is never scanned)
    for (;;) {
        PF_LAP(13)
        PF_ADD(6, (!have && nx) ? 1u : 0u)
        // ---- a lane without a node takes up the one that waits in its slot (it arrived with the previous round's memory step)
        if (!have && nx) {
            const uint4 n0 = ldsNext[0][tid], n1 = ldsNext[1][tid], n2 = ldsNext[2][tid];
            nx = false;
            have = true;
            fresh = true;
            parent = RangePair{{n0.x, n0.y}, {n0.z, n0.w}};
            row = n1.x & 0xFFFFu;
            ctx = n1.y;
            fcP = n1.z;
            md = (int)((n1.w >> 8) & 3u);
            blk = (row + 1u) / Geo::BLOCK;
            curIdx = nxIdx;
            pA[cur][tid] = n2;
            pB[cur][tid] = make_uint2(n1.x, n1.w & 0x3FFu);
        }
        CMB_LDS_SYNC(); // (the slot is read before it is requested again)
        // ---- ask for the next one: the lanes without a waiting node share out the next entries of the wavefront's chunks
        uint32_t n2Load;
        {
            const bool want = !nx && chunk < nChunks;
            const unsigned long long req = __ballot(want);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(req >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)req, 0u));
            const uint32_t e = cUsed + rank;
            uint32_t idx = 0xFFFFFFFFu;
            if (e < cCnt) idx = chunk * 64u + e;
            else if (e - cCnt < nCnt) idx = (chunk + W) * 64u + (e - cCnt);
            if (want && idx != 0xFFFFFFFFu) {
                gldsU4(Qi + idx, &ldsNext[0][wbase]);
                gldsU4(Qi + (size_t)qCap + idx, &ldsNext[1][wbase]);
                gldsU4(Qi + (size_t)2 * qCap + idx, &ldsNext[2][wbase]);
                nx = true;
                nxIdx = idx;
            }
            cUsed += (uint32_t)__popcll(req);
            if (cUsed >= cCnt && chunk < nChunks) { // this chunk is handed out: on to the next (the counts are here already)
                cUsed = min(cUsed - cCnt, nCnt);
                chunk += W;
                cCnt = nCnt;
                nCnt = chunk + W < nChunks ? n2 : 0u;
            }
            n2Load = cntIn[min(chunk + 2u * W, cntLast)]; // (arrives with this round's memory step)
        }
        const bool exhausted = chunk >= nChunks; // (wave-uniform)
        if (__ballot(have || nx) == 0ull && exhausted) break;
        const bool walkOk = !exhausted || tail != 0u;
        if (exhausted && tail) tail--;
        PF_LAP(8)
        // ---- the memory step of the round: rank blocks of every lane that holds a node; hot word, match words and final-column
        // pack of a fresh node; match words of a walk that enters a new row block
        uint4 rk[4];
        uint4 hotN = make_uint4(0, 0, 0, 0);
        const bool hasPack = have && fresh && fcP != BFS_NONE; // (only a fresh node can lie in the final column: such nodes are not walked on with)
        if (have) {
            issueRanks(ix, md, parent, rk);
            if (fresh || needM) {
                const uint4* Cx = B.C + (size_t)CMB_IDX(ctx, B.cCap, 1) * B.ctxU4;
                if (fresh) hotN = Cx[CTX_HOT];
                gldsU4(Cx + CTX_M + 2u * blk, &ldsM4[0][wbase]);
                gldsU4(Cx + CTX_M + 1u + 2u * blk, &ldsM4[1][wbase]);
                if (hasPack) {
#pragma unroll
                    for (uint32_t u = 0; u < PK; u++) gldsU4(Qi + (size_t)(3u + u) * qCap + curIdx, &ldsPk[u][wbase]);
                }
            }
        }
        PF_LAP(9)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PF_LAP(2)
        PF_ADD(0, (tid & 63u) == 0 ? 1u : 0u)
        PF_ADD(1, have ? 1u : 0u)
        n2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(n2Load, 64u));
        // ---- parents: the four child ranges, the non-empty ones onto the wavefront's child list
        uint32_t nKids = 0;
        RangePair kid[4];
        uint32_t kidMask = 0;
        if (have) {
            if (fresh) {
                pG[tid] = hotN;
                pF[tid] = make_uint4(ctx, fcP, hasPack ? 1u : 0u, 0u);
                hotY = hotN.y;
                clSize = hotN.w >> 23;
            }
            fresh = false;
            needM = false;
            k4[tid] = 0u;
            cntExp++;
            uint32_t Rb[4], Re[4], db, de;
            takeRanks(ix, md, parent, rk, Rb, Re, db, de);
#pragma unroll
            for (uint32_t ch = 1; ch <= 4; ch++) {
                const bool ne = childFromRanks(ix, md, parent, ch, Rb, Re, db, de, kid[ch - 1]);
                kidMask |= (ne ? 1u : 0u) << (ch - 1);
            }
            nKids = (uint32_t)__popc(kidMask);
            cntChildren += nKids;
            { // (a row beyond the matrix)
                const uint32_t m = (hotY >> 9) & 0x1FFu, sfc = ((hotY >> 23) & 15u) + ((hotY >> 18) & 31u) + 1u;
                if (row + 1u >= m - sfc && clSize + row + 1u - m >= Geo::CELLS) flags |= FLAG_CAPACITY;
            }
        }
        const uint32_t incl = waveInclusiveScanDpp(nKids);
        const uint32_t T = waveLastLane(incl); // children of the wavefront in this round
        if (have) {
            uint32_t o = incl - nKids;
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                if ((kidMask >> c) & 1u) {
                    clR[wv][o] = make_uint4(kid[c].sa.b, kid[c].sa.e, kid[c].rev.b, kid[c].rev.e);
                    clT[wv][o] = (uint8_t)(lane | (c << 6));
                    o++;
                }
            }
        }
        CMB_LDS_SYNC();
        PF_LAP(10)
        PF_ADD(7, nKids)
        // ---- children: one per lane, 64 at a time
        for (uint32_t g0 = 0; g0 < T; g0 += 64u) {
            PF_ADD(4, (tid & 63u) == 0 ? 1u : 0u)
            const uint32_t g = g0 + lane;
            uint32_t res = KIND_NONE, par = tid, ch = 1, row1 = 0, mdC = 0;
            ChildState cs{};
            uint4 hg = make_uint4(0, 0, 0, 0);
            if (g < T) {
                const uint32_t t = clT[wv][g];
                par = wbase + (t & 63u);
                ch = (t >> 6) + 1u;
                const uint4 r = clR[wv][g];
                const uint4 a = pA[cur][par];
                const uint2 b = pB[cur][par];
                hg = pG[par];
                const uint64_t M = reinterpret_cast<const uint64_t*>(&ldsM4[(ch - 1u) >> 1][par])[(ch - 1u) & 1u];
                ExpandCtx e{};
                e.switchPoint = ix.switchPoint;
                e.itStart = hg.z;
                e.g.n = hg.y & 0x1FFu;
                e.g.m = (hg.y >> 9) & 0x1FFu;
                e.g.Wv = (hg.y >> 18) & 31u;
                e.g.Wh = (hg.y >> 23) & 15u;
                e.g.maxED = (hg.y >> 27) & 15u;
                e.clSize = hg.w >> 23;
                e.itMode = (hg.x >> 25) & 3u; // 0: phase 0 (no switch), 1: start difference fixed, 2: BACKWARD
                row1 = (b.x & 0xFFFFu) + 1u;
                mdC = (b.y >> 8) & 3u;
                const bool inFC = e.g.inFinalColumn(row1);
                res = evalRow<Geo>(RangePair{{r.x, r.y}, {r.z, r.w}}, e, row1, inFC, M, u64of(a.x, a.y), u64of(a.z, a.w), 1ull << (b.y & 63u),
                                   b.x >> 16, cs);
                if (res & 8u) flags |= FLAG_CAPACITY;
            }
            uint32_t kd = res & 3u;
            const bool wantF = (res >> 2) & 1u;
            // the first plain node among a parent's children claims the parent's lane: the walk goes on with it
            if (kd == KIND_NODE && !wantF && walkOk) {
                const uint32_t old = atomicOr(&k4[par], 1u);
                if (!(old & 1u)) {
                    aR[par] = make_uint4(cs.r.sa.b, cs.r.sa.e, cs.r.rev.b, cs.r.rev.e);
                    pA[cur ^ 1u][par] = make_uint4((uint32_t)cs.HP, (uint32_t)(cs.HP >> 32), (uint32_t)cs.HN, (uint32_t)(cs.HN >> 32));
                    pB[cur ^ 1u][par] = make_uint2(row1 | (cs.sc << 16), ((uint32_t)__ffsll((unsigned long long)cs.RAC) - 1u) | (mdC << 8));
                    kd = KIND_NONE;
                }
            }
            // ---- slots for what leaves (one prefix sum over the child lanes for nodes, events and F records, one for the items)
            const uint32_t nNode = kd == KIND_NODE ? 1u : 0u, nEv = kd == KIND_EVENT ? 1u : 0u, nF = (kd != KIND_NONE && wantF) ? 1u : 0u;
            const uint32_t nIt = kd == KIND_ITEMS ? cs.r.sa.e - cs.r.sa.b : 0u;
            const uint32_t pk3 = nNode | (nEv << 8) | (nF << 16);
            const uint32_t in3 = waveInclusiveScanDpp(pk3), t3 = waveLastLane(in3);
            const uint32_t inI = waveInclusiveScanDpp(nIt), tI = waveLastLane(inI);
            bool ovQ = false, ovE = false, ovI = false, ovF = false;
            uint32_t oNode = 0xFFFFFFFFu;
            if (t3 & 0xFFu) oNode = ncNode.alloc(&B.nq[pass + 1], qCap, t3 & 0xFFu, regionSlots, cntOut, ovQ);
            const uint32_t oEv = wcEv.allocPre(&B.ne[pass + 1], B.evCap, ((in3 >> 8) & 0xFFu) - nEv, (t3 >> 8) & 0xFFu, chEv, ovE, holeEv);
            const uint32_t oF = wcF.allocPre(&B.pool[0], B.fCap, ((in3 >> 16) & 0xFFu) - nF, (t3 >> 16) & 0xFFu, WALK_CH_F, ovF, holeF);
            const uint32_t oIt = wcIt.allocPre(&q.cnt[0], q.itemCap, inI - nIt, tI, WALK_CH_IT, ovI, holeIt);
            if (oNode != 0xFFFFFFFFu) oNode += (in3 & 0xFFu) - nNode;
            if (ovQ) flags |= FLAG_BFS_Q;
            if (ovE) flags |= FLAG_BFS_EV;
            if (ovI) flags |= FLAG_ITEM_OVERFLOW;
            if (ovF) flags |= FLAG_BFS_F;
            // (a wavefront whose share does not fit drops it: the host sees the needed sizes and re-runs)
            if (kd != KIND_NONE && !(ovQ || ovE || ovI || ovF)) {
                const uint4 pf = pF[par];
                const uint32_t ctxC = pf.x, fcPC = pf.y;
                const uint32_t mC = (hg.y >> 9) & 0x1FFu;
                const uint32_t cell = min((hg.w >> 23) + row1 - mC, Geo::CELLS - 1u);
                typename Geo::Pack pack{}; // final-column distances of the path so far
                if (pf.z && (wantF || kd == KIND_EVENT)) packLoad(&ldsPk[0][par], 256, pack);
                const uint4 cr = make_uint4(cs.r.sa.b, cs.r.sa.e, cs.r.rev.b, cs.r.rev.e);
                uint32_t fc = BFS_NONE;
                if (wantF) {
                    fc = oF;
                    uint4* Fr = B.F + (size_t)CMB_IDX(fc, B.fCap, 9) * F_U4;
                    Fr[0] = cr;
                    Fr[1] = make_uint4(row1 | (ch << 16), fcPC, 0u, 0u);
                }
                if (kd == KIND_NODE) {
                    const uint32_t o = oNode;
                    Qo[o] = cr;
                    Qo[(size_t)qCap + o] = make_uint4(row1 | (cs.sc << 16), ctxC, fc,
                                                      ((uint32_t)__ffsll((unsigned long long)cs.RAC) - 1u) | (mdC << 8));
                    Qo[(size_t)2 * qCap + o] = make_uint4((uint32_t)cs.HP, (uint32_t)(cs.HP >> 32), (uint32_t)cs.HN,
                                                          (uint32_t)(cs.HN >> 32));
                    if (wantF) {
                        typename Geo::Pack p2 = pack;
                        edPut(p2, cell, cs.aux);
                        packStore(Qo + (size_t)3 * qCap + o, qCap, p2);
                    }
                } else if (kd == KIND_EVENT) {
                    typename Geo::Pack p2 = pack;
                    edPut(p2, cell, cs.aux);
                    Eo[(size_t)EV_U4 * oEv] = make_uint4(ctxC, fc, 0xFFFFFFFFu, cell);
                    packStore(Eo + (size_t)EV_U4 * oEv + 1, 1, p2);
                } else {
                    const uint32_t w = cs.r.sa.e - cs.r.sa.b;
                    const uint32_t rsId = hg.x & 0x1FFFFFFu, itMeta = hg.w & 0x7FFFFFu;
                    for (uint32_t t = 0; t < w; t++) q.items[oIt + t] = make_uint4(rsId, cs.r.sa.b + t, cs.aux, itMeta);
                }
            }
        }
        CMB_LDS_SYNC();
        PF_LAP(11)
        // ---- parents again: walk on with the child that claimed the lane, or let go of the node
        if (have) {
            if (k4[tid] & 1u) {
                const uint4 r = aR[tid];
                parent = RangePair{{r.x, r.y}, {r.z, r.w}};
                row = row + 1u;
                if ((row + 1u) / Geo::BLOCK != blk) { // the next row lies in another row block: its match words
                    blk = (row + 1u) / Geo::BLOCK;
                    needM = true;
                }
            } else {
                have = false;
            }
        }
        cur ^= 1u;
        PF_ADD(5, have ? 1u : 0u)
        PF_LAP(12)
    }
#ifdef CMB_BFS_STATS
    if ((tid & 63u) == 0) pf[3] = (unsigned long long)(clock64() - pfT0);
    for (int j = 0; j < 16; j++) {
        unsigned long long x = pf[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        if ((tid & 63u) == 0 && x) atomicAdd(&g_bfsStats[j], x);
    }
#endif
    ncNode.retire(cntOut);
    wcEv.fill(holeEv);
    if ((tid & 63u) == 0) { // the item and F chunks go on in the wavefront's next pass
        save[0] = wcIt.base, save[1] = wcIt.used, save[2] = wcIt.size;
        save[4] = wcF.base, save[5] = wcF.used, save[6] = wcF.size;
    }
    // per-block counters (summed by k_bfs_finish): one writer per slot and launch, launches are ordered
    unsigned long long v[3] = {cntChildren, cntExp, cntChildren}; // (every child gets its matrix row)
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

// ------------------------------------------------------------------ expand: bfsExpand per WAVEFRONT (round 4, CMB_BFS_WALK=2)
// The per-node logic of bfsExpand, lane for lane (same classification, chain walk and records), without what a block-wide tile costs
// it: three barriers and an atomic round trip per tile, and the node planes as a round trip of their own.  A wavefront works on its
// own chunks of the chunked node queue (chunk c belongs to wavefront c % W, counts in BfsBufs::qCnt — as bfsExpandWalk):
//   * the node planes of the NEXT chunk are copied into the lanes' LDS slots by global_load_lds while the current one is expanded;
//   * queue slots come from per-wavefront chunks: one prefix sum over the lanes (DPP), an atomic only when a chunk is used up;
//   * no barrier: the four wavefronts of a block drift apart, which is what hides their memory round trips from each other.
template <class Geo = GeoN>
__device__ __forceinline__ void bfsExpandWave(const DevIndex& ix, const BfsBufs& B, uint32_t pass, const Queues& q, uint32_t bid,
                                              uint32_t nBlocks) {
    constexpr uint32_t EV_U4 = 1u + Geo::PK_U4;
    __shared__ uint4 ldsNext[3][256];  // node planes of the wavefront's next chunk
    __shared__ uint64_t ldsM[4][256];
    __shared__ uint32_t ldsR[8][256];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wbase = tid & ~63u;
    const uint32_t nIn = min(B.nq[pass], B.qCap);
    const uint4* __restrict__ Qi = B.Q[pass & 1u];
    uint4* __restrict__ Qo = B.Q[(pass + 1u) & 1u];
    uint4* __restrict__ Eo = B.Ev[(pass + 1u) & 1u];
    const uint32_t* __restrict__ cntIn = B.qCnt[pass & 1u];
    uint32_t* __restrict__ cntOut = B.qCnt[(pass + 1u) & 1u];
    const uint32_t qCap = B.qCap;
    const uint32_t W = nBlocks * 4u, wId = bid * 4u + (tid >> 6), nChunks = (nIn + 63u) >> 6;
    const uint32_t cntLast = nChunks ? nChunks - 1u : 0u;
    uint32_t chunk = wId;
    uint32_t cCnt = 0, nCnt = 0;
    {
        const uint32_t c0 = cntIn[min(chunk, cntLast)], c1 = cntIn[min(chunk + W, cntLast)];
        cCnt = chunk < nChunks ? (uint32_t)__builtin_amdgcn_readfirstlane((int)min(c0, 64u)) : 0u;
        nCnt = chunk + W < nChunks ? (uint32_t)__builtin_amdgcn_readfirstlane((int)min(c1, 64u)) : 0u;
        if (lane < cCnt) {
            const uint32_t i0 = chunk * 64u + lane;
            gldsU4(Qi + i0, &ldsNext[0][wbase]);
            gldsU4(Qi + (size_t)qCap + i0, &ldsNext[1][wbase]);
            gldsU4(Qi + (size_t)2 * qCap + i0, &ldsNext[2][wbase]);
        }
    }
    uint32_t flags = 0, cntChildren = 0, cntExp = 0;
    NodeRegion ncNode;
    const uint32_t regionSlots = nodeRegionSlots(nIn, W);
    WaveChunk wcEv, wcIt, wcF;
    uint32_t* save = B.wcSave + (size_t)wId * WALK_SAVE_U32;
    { // the item and F chunks of the wavefront's previous pass (zeroed before the search)
        wcIt.base = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[0]);
        wcIt.used = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[1]);
        wcIt.size = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[2]);
        wcF.base = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[4]);
        wcF.used = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[5]);
        wcF.size = (uint32_t)__builtin_amdgcn_readfirstlane((int)save[6]);
    }
    const uint32_t chEv = nIn < 262144u ? WALK_CH_EV_SMALL : WALK_CH_EV;
    auto holeEv = [&](uint32_t o) { Eo[(size_t)EV_U4 * o] = make_uint4(BFS_NONE, 0u, 0u, 0u); };
    auto holeIt = [&](uint32_t o) { q.items[o] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u); };
    auto holeF = [&](uint32_t) {};
#ifdef CMB_BFS_STATS
    // diagnostic build only (tools/walk_stats.sh): [0] tiles, [1] active lanes, [3] cycles in the loop, [8..13] cycles: wait for the
    // planes | first memory step + classification | chain steps | allocation | output | loop end
    unsigned long long pf[16] = {};
    const long long pfT0 = clock64();
    long long pfT = pfT0;
#endif
    while (chunk < nChunks) { // (wave-uniform)
        PF_LAP(13)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the chunk's node planes are in the lanes' slots
        PF_LAP(8)
        PF_ADD(0, lane == 0 ? 1u : 0u)
        PF_ADD(1, lane < cCnt ? 1u : 0u)
        const bool act = lane < cCnt;
        const uint32_t i = chunk * 64u + lane;
        uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0, n2 = n0;
        if (act) {
            n0 = ldsNext[0][tid];
            n1 = ldsNext[1][tid];
            n2 = ldsNext[2][tid];
        }
        CMB_LDS_SYNC(); // (the slots are read before they are filled again)
        // the next chunk of the wavefront: its planes travel while this one is expanded; the count of the one after it
        const uint32_t chunkN = chunk + W;
        if (chunkN < nChunks && lane < nCnt) {
            const uint32_t i1 = chunkN * 64u + lane;
            gldsU4(Qi + i1, &ldsNext[0][wbase]);
            gldsU4(Qi + (size_t)qCap + i1, &ldsNext[1][wbase]);
            gldsU4(Qi + (size_t)2 * qCap + i1, &ldsNext[2][wbase]);
        }
        const uint32_t n2Load = cntIn[min(chunkN + W, cntLast)];
        if (cCnt == 0u) { // an empty chunk (the unused rest of a producer's region): on to the next
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            chunk = chunkN;
            cCnt = nCnt;
            nCnt = chunk + W < nChunks ? (uint32_t)__builtin_amdgcn_readfirstlane((int)min(n2Load, 64u)) : 0u;
            continue;
        }
        uint32_t kinds = 0; // 4 bits per child: kind | needF << 2
        uint32_t row1 = 0, ctx = 0, fcP = BFS_NONE, nIt = 0;
        RangePair parent{{0, 0}, {0, 0}};
        uint32_t row = 0, score = 0, blk = 0;
        int md = 0;
        uint64_t pHP = 0, pHN = 0;
        uint32_t pRac = 0;
        ExpandCtx e{};
        e.switchPoint = ix.switchPoint;
        uint32_t db = 0, de = 0;
        uint32_t hotX = 0, hotW = 0;
        bool walking = act;
        if (act) {
            ctx = n1.y;
            fcP = n1.z;
            row = n1.x & 0xFFFFu;
            score = n1.x >> 16;
            md = (int)((n1.w >> 8) & 3u);
            parent = RangePair{{n0.x, n0.y}, {n0.z, n0.w}};
            uint4 rk[4];
            issueRanks(ix, md, parent, rk);
            const uint4* Cx = B.C + (size_t)CMB_IDX(ctx, B.cCap, 1) * B.ctxU4;
            blk = (row + 1) / Geo::BLOCK;
            const uint4 hot = Cx[CTX_HOT];
            const uint4 mA = Cx[CTX_M + 2 * blk], mB = Cx[CTX_M + 1 + 2 * blk];
            {
                uint32_t Rb[4], Re[4];
                takeRanks(ix, md, parent, rk, Rb, Re, db, de);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    ldsR[c][tid] = Rb[c];
                    ldsR[4 + c][tid] = Re[c];
                }
            }
            e.itStart = hot.z;
            e.g.n = hot.y & 0x1FFu;
            e.g.m = (hot.y >> 9) & 0x1FFu;
            e.g.Wv = (hot.y >> 18) & 31u;
            e.g.Wh = (hot.y >> 23) & 15u;
            e.g.maxED = (hot.y >> 27) & 15u;
            e.clSize = hot.w >> 23;
            e.itMode = (hot.x >> 25) & 3u;
            hotX = hot.x;
            hotW = hot.w;
            pHP = u64of(n2.x, n2.y);
            pHN = u64of(n2.z, n2.w);
            pRac = n1.w & 63u;
            ldsM[0][tid] = u64of(mA.x, mA.y);
            ldsM[1][tid] = u64of(mA.z, mA.w);
            ldsM[2][tid] = u64of(mB.x, mB.y);
            ldsM[3][tid] = u64of(mB.z, mB.w);
        }
        // ---- walk (as bfsExpand)
        for (uint32_t step = 0; step < B.chain; step++) { // (wave-uniform exit below)
            if (step == 1u) { PF_LAP(9) }
            if (walking) {
                if (step) {
                    uint4 rk[4];
                    issueRanks(ix, md, parent, rk);
                    uint32_t Rb[4], Re[4];
                    takeRanks(ix, md, parent, rk, Rb, Re, db, de);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        ldsR[c][tid] = Rb[c];
                        ldsR[4 + c][tid] = Re[c];
                    }
                }
                uint32_t Rb[4], Re[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    Rb[c] = ldsR[c][tid];
                    Re[c] = ldsR[4 + c][tid];
                }
                row1 = row + 1;
                cntExp++;
                const bool inFC = e.g.inFinalColumn(row1);
                if (inFC && e.clSize + row1 - e.g.m >= Geo::CELLS) flags |= FLAG_CAPACITY;
                kinds = 0;
                nIt = 0;
                uint32_t nOut = 0, nChildren = 0;
#pragma unroll
                for (uint32_t ch = 1; ch <= 4; ch++) {
                    bool nonEmpty;
                    ChildState cs;
                    const uint32_t k4 = evalChild<false, Geo>(ix, md, parent, ch, Rb, Re, db, de, e, row1, inFC, ldsM[ch - 1][tid], pHP, pHN,
                                                              1ull << pRac, score, nonEmpty, cs);
                    nChildren += nonEmpty ? 1u : 0u;
                    if (k4 & 8u) flags |= FLAG_CAPACITY;
                    if ((k4 & 3u) == KIND_NONE) continue;
                    kinds |= (k4 & 7u) << (4 * (ch - 1));
                    nOut++;
                    if ((k4 & 3u) == KIND_ITEMS) {
                        RangePair child;
                        (void)childFromRanks(ix, md, parent, ch, Rb, Re, db, de, child);
                        nIt += child.sa.e - child.sa.b;
                    }
                }
                cntChildren += nChildren;
                const bool single = nOut == 1u && (kinds == 0x1u || kinds == 0x10u || kinds == 0x100u || kinds == 0x1000u);
                if (single && step + 1u < B.chain && (row1 + 1u) / Geo::BLOCK == blk) {
                    const uint32_t ch = ((31u - (uint32_t)__clz(kinds)) >> 2) + 1u;
                    const uint64_t M = ch == 1 ? ldsM[0][tid] : ch == 2 ? ldsM[1][tid] : ch == 3 ? ldsM[2][tid] : ldsM[3][tid];
                    bool nonEmpty;
                    ChildState one;
                    (void)evalChild<true, Geo>(ix, md, parent, ch, Rb, Re, db, de, e, row1, inFC, M, pHP, pHN, 1ull << pRac, score, nonEmpty, one);
                    parent = one.r;
                    score = one.sc;
                    pHP = one.HP;
                    pHN = one.HN;
                    pRac = (uint32_t)__ffsll((unsigned long long)one.RAC) - 1u;
                    row = row1;
                    kinds = 0;
                } else {
                    walking = false;
                }
            }
            if (__ballot(walking) == 0ull) break;
        }
        PF_LAP(10)
        // ---- slots: one prefix sum over the lanes for nodes, events and F records, one for the items
        uint32_t nNode = 0, nEv = 0, nF = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t kd = (kinds >> (4 * c)) & 3u;
            nNode += kd == KIND_NODE;
            nEv += kd == KIND_EVENT;
            nF += (kinds >> (4 * c + 2)) & 1u;
        }
        const uint32_t pk3 = nNode | (nEv << 10) | (nF << 20);
        const uint32_t in3 = waveInclusiveScanDpp(pk3), t3 = waveLastLane(in3);
        const uint32_t inI = waveInclusiveScanDpp(nIt), tI = waveLastLane(inI);
        bool ovE = false, ovI = false, ovF = false;
        bool ovQ = false;
        const uint32_t nb0 = (t3 & 0x3FFu) ? ncNode.alloc(&B.nq[pass + 1], qCap, t3 & 0x3FFu, regionSlots, cntOut, ovQ) : 0u;
        const bool okQ = !ovQ;
        uint32_t oEv = wcEv.allocPre(&B.ne[pass + 1], B.evCap, ((in3 >> 10) & 0x3FFu) - nEv, (t3 >> 10) & 0x3FFu, chEv, ovE, holeEv);
        uint32_t oF = wcF.allocPre(&B.pool[0], B.fCap, ((in3 >> 20) & 0x3FFu) - nF, (t3 >> 20) & 0x3FFu, WALK_CH_F, ovF, holeF);
        uint32_t oIt = wcIt.allocPre(&q.cnt[0], q.itemCap, inI - nIt, tI, WALK_CH_IT, ovI, holeIt);
        uint32_t pNode = (in3 & 0x3FFu) - nNode; // this lane's first node among the wavefront's
        if (!okQ) flags |= FLAG_BFS_Q;
        if (ovE) flags |= FLAG_BFS_EV;
        if (ovI) flags |= FLAG_ITEM_OVERFLOW;
        if (ovF) flags |= FLAG_BFS_F;
        PF_LAP(11)
        if (kinds != 0u && okQ && !ovE && !ovI && !ovF) {
            const bool inFC = e.g.inFinalColumn(row1);
            const uint32_t cell = min(e.clSize + row1 - e.g.m, Geo::CELLS - 1u);
            typename Geo::Pack pack{};
            if (fcP != BFS_NONE && (kinds & 0x4444u)) packLoad(Qi + (size_t)3 * qCap + i, qCap, pack);
            const uint32_t rsId = hotX & 0x1FFFFFFu, itMeta = hotW & 0x7FFFFFu;
            uint32_t Rb[4], Re[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                Rb[c] = ldsR[c][tid];
                Re[c] = ldsR[4 + c][tid];
            }
#pragma unroll
            for (uint32_t ch = 1; ch <= 4; ch++) {
                const uint32_t kd = (kinds >> (4 * (ch - 1))) & 3u;
                if (kd == KIND_NONE) continue;
                bool nonEmpty;
                ChildState cs;
                (void)evalChild<true, Geo>(ix, md, parent, ch, Rb, Re, db, de, e, row1, inFC, ldsM[ch - 1][tid], pHP, pHN, 1ull << pRac, score,
                                           nonEmpty, cs);
                const bool wantF = (kinds >> (4 * (ch - 1) + 2)) & 1u;
                const uint4 cr = make_uint4(cs.r.sa.b, cs.r.sa.e, cs.r.rev.b, cs.r.rev.e);
                uint32_t fc = BFS_NONE;
                if (wantF) {
                    fc = oF++;
                    uint4* Fr = B.F + (size_t)CMB_IDX(fc, B.fCap, 9) * F_U4;
                    Fr[0] = cr;
                    Fr[1] = make_uint4(row1 | (ch << 16), fcP, 0u, 0u);
                }
                if (kd == KIND_NODE) {
                    const uint32_t o = nb0 + pNode;
                    pNode++;
                    Qo[o] = cr;
                    Qo[(size_t)qCap + o] = make_uint4(row1 | (cs.sc << 16), ctx, fc,
                                                      ((uint32_t)__ffsll((unsigned long long)cs.RAC) - 1u) | ((uint32_t)md << 8));
                    Qo[(size_t)2 * qCap + o] = make_uint4((uint32_t)cs.HP, (uint32_t)(cs.HP >> 32), (uint32_t)cs.HN, (uint32_t)(cs.HN >> 32));
                    if (wantF) {
                        typename Geo::Pack p2 = pack;
                        edPut(p2, cell, cs.aux);
                        packStore(Qo + (size_t)3 * qCap + o, qCap, p2);
                    }
                } else if (kd == KIND_EVENT) {
                    typename Geo::Pack p2 = pack;
                    edPut(p2, cell, cs.aux);
                    Eo[(size_t)EV_U4 * oEv] = make_uint4(ctx, fc, 0xFFFFFFFFu, cell);
                    packStore(Eo + (size_t)EV_U4 * oEv + 1, 1, p2);
                    oEv++;
                } else {
                    const uint32_t w = cs.r.sa.e - cs.r.sa.b;
                    for (uint32_t t = 0; t < w; t++) q.items[oIt + t] = make_uint4(rsId, cs.r.sa.b + t, cs.aux, itMeta);
                    oIt += w;
                }
            }
        }
        PF_LAP(12)
        // ---- on to the wavefront's next chunk
        chunk = chunkN;
        cCnt = nCnt;
        nCnt = chunk + W < nChunks ? (uint32_t)__builtin_amdgcn_readfirstlane((int)min(n2Load, 64u)) : 0u;
    }
#ifdef CMB_BFS_STATS
    if (lane == 0) pf[3] = (unsigned long long)(clock64() - pfT0);
    for (int j = 0; j < 16; j++) {
        unsigned long long x = pf[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        if (lane == 0 && x) atomicAdd(&g_bfsStats[j], x);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ncNode.retire(cntOut);
    wcEv.fill(holeEv);
    if (lane == 0) { // the item and F chunks go on in the wavefront's next pass
        save[0] = wcIt.base, save[1] = wcIt.used, save[2] = wcIt.size;
        save[4] = wcF.base, save[5] = wcF.used, save[6] = wcF.size;
    }
    unsigned long long v[3] = {cntChildren, cntExp, cntChildren};
#pragma unroll
    for (int j = 0; j < 3; j++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[j] += __shfl_xor(v[j], d);
    }
    __shared__ unsigned long long shc[4][3];
    if (lane == 0)
        for (int j = 0; j < 3; j++) shc[tid >> 6][j] = v[j];
    __syncthreads();
    if (tid < 3) {
        const unsigned long long t = shc[0][tid] + shc[1][tid] + shc[2][tid] + shc[3][tid];
        if (t) B.blockCnt[(size_t)bid * 4 + tid] += t;
    }
    if (flags) atomicOr(&q.cnt[3], flags);
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

template <bool START, class Tr = FmTraits, class Geo = GeoN>
__device__ __forceinline__ void bfsHeavy(const DevStrategyKT<Geo::MP>* __restrict__ stp, const BfsBufs& B,
                                         uint32_t pass, const typename Tr::Task* __restrict__ tasks, uint32_t nTasks,
                                         const uint64_t* __restrict__ offs, uint32_t gw, const uint32_t* __restrict__ G,
                                         const PartOutT<Geo::MP>* __restrict__ parts, const Queues& q, uint32_t bid,
                                         uint32_t nBlocks) {
    typedef DevSearchT<Geo::MP> DevSearch; // (the instance's table size)
    typedef PartOutT<Geo::MP> PartOut;
    typedef typename Geo::Pack EdPack;     // ... and final-column pack
    constexpr uint32_t ED_CELLS = Geo::CELLS, ED_MAX = Geo::ED_MAX, PK = Geo::PK_U4, EV_U4 = 1u + Geo::PK_U4;
    __shared__ uint32_t sh[4][5];
    __shared__ uint8_t ieL[ED_CELLS + 2][256]; // initEds under construction, [entry][thread]
    typedef typename Tr::Pair Pair;
    constexpr uint32_t PU = Tr::PAIR_U4;  // uint4 per range pair
    constexpr uint32_t FU = PU + 1;       // F record: pair, {depth | c << 16, parent, reported, -}
    constexpr uint32_t DU = PU + 1;       // descendant of a list: pair, {depth | c << 16}
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
        EdPack pack{};
        Pair startR = Tr::none();
        uint32_t startDepth = 0;
        const DevSearch* s = nullptr;
        if (i < nIn) {
            if (START) {
                const typename Tr::Task t = tasks[i];
                if (t.rsId != 0xFFFFFFFFu) { // (holes of the task queue)
                    P.kind = 4;
                    rsId = t.rsId;
                    scheme = t.scheme;
                    search = t.search;
                    idx = t.idx; // the phase to enter
                    startR = Tr::taskRange(t);
                    startDepth = t.depth;
                    s = &stp->sch[scheme].s[search];
                }
            } else if (Ei[(size_t)EV_U4 * i].x != BFS_NONE) { // (holes: the unused rest of a wavefront's chunk, bfsExpandWalk)
                const uint4 ev = Ei[(size_t)EV_U4 * i];
                c0i = ev.x;
                fcE = ev.y;
                remFrom = (int)ev.z;
                last = ev.w;
                const uint4* Cx = B.C + (size_t)CMB_IDX(c0i, B.cCap, 3) * B.ctxU4;
                const uint4 c0 = Cx[0], c1 = Cx[1], c3 = Cx[3];
                packLoad(Ei + (size_t)EV_U4 * i + 1, 1, pack); // final-column distances of the path (travel with the event)
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
                            const uint32_t dn = B.C[(size_t)CMB_IDX(descRef0, B.cCap, 4) * B.ctxU4 + 4].y & 0xFFu;
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
                dC4 = B.C[(size_t)CMB_IDX(descRefN, B.cCap, 5) * B.ctxU4 + 4];
                P.nDescSrc = dC4.y & 0xFFu;
            }
        }
        // ---- block-wide allocation: contexts, F records, arena units, in-index occurrences
        const uint32_t wantCtx = P.kind >= 2 ? 1u : 0u;
        const uint32_t wantF = P.kind >= 2 ? 1u + P.nDescSrc : 0u;
        const uint32_t wantA = P.kind == 3 ? DU * P.nDescNew + (2u * P.ni + 15u) / 16u : 0u;
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
        Pair smR = Tr::none();
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
                const uint4 f1 = B.F[(size_t)CMB_IDX(cur, B.fCap, 10) * FU + PU];
                if ((P.centres >> c) & 1u) {
                    const uint32_t old = atomicExch(&reinterpret_cast<uint32_t*>(B.F + (size_t)CMB_IDX(cur, B.fCap, 11) * FU + PU)[2], 1u);
                    if (!old) { // FMPosExt::report (indexhelpers.h:1586-1601): once per node
                        const Pair r = Tr::load(B.F + (size_t)CMB_IDX(cur, B.fCap, 12) * FU, 1);
                        const uint32_t e = edGet(pack, c);
                        if (!Tr::empty(r) && e >= lowerBound)
                            Tr::emitFm(B, q, fmNext++, rsId, r, (f1.x & 0xFFFFu) + smDepth, e, smShift);
                    }
                }
                if (c == lowest) break;
                cur = f1.y;
            }
            for (; fmNext < fmEnd; fmNext++) Tr::fmHole(B, q, fmNext); // holes
        } else if (P.kind == 2) {
            uint32_t cur = fcE;
            for (uint32_t c = last; c > P.ci; c--) cur = B.F[(size_t)CMB_IDX(cur, B.fCap, 13) * FU + PU].y;
            const uint32_t old = atomicExch(&reinterpret_cast<uint32_t*>(B.F + (size_t)CMB_IDX(cur, B.fCap, 14) * FU + PU)[2], 1u);
            if (!old) {
                const uint4 f1 = B.F[(size_t)CMB_IDX(cur, B.fCap, 16) * FU + PU];
                const uint32_t up = P.ci - P.hi;
                smR = Tr::load(B.F + (size_t)CMB_IDX(cur, B.fCap, 15) * FU, 1);
                smDist = P.ed;
                smDepthN = (f1.x & 0xFFFFu) + (smDepth - up);
                smShiftN = (dirCur == 1 ? up : 0u) + smShift;
                enter = !Tr::empty(smR) && smDist >= lowerBound;
            }
        } else if (P.kind == 3) {
            // descendants = the final-column nodes below the centre (walked bottom-up), then the rest of the
            // interrupted replay; depths renumbered 1.. (:627-630)
            uint4* dl = B.A + CMB_IDX(aOff, B.aCap, 101);
            uint32_t cur = fcE;
            for (uint32_t c = last; c > P.ci; c--) {
                const uint4* Fc = B.F + (size_t)CMB_IDX(cur, B.fCap, 17) * FU;
                const uint4 f1 = Fc[PU];
                const uint32_t j = c - P.ci - 1;
#pragma unroll
                for (uint32_t u = 0; u < PU; u++) dl[DU * j + u] = Fc[u];
                dl[DU * j + PU] = make_uint4((j + 1) | (f1.x & 0xFF0000u), 0u, 0u, 0u);
                cur = f1.y;
            }
            const uint4 f1 = B.F[(size_t)CMB_IDX(cur, B.fCap, 20) * FU + PU];
            smR = Tr::load(B.F + (size_t)CMB_IDX(cur, B.fCap, 19) * FU, 1);
            smDist = P.ed;
            smDepthN = (f1.x & 0xFFFFu) + smDepth;
            smShiftN = smShift;
            enter = !Tr::empty(smR);
            if (enter) {
                const uint32_t nd0 = last - P.ci;
                if (P.nRem) {
                    const uint4 sC4 = B.C[(size_t)CMB_IDX(descRef0, B.cCap, 6) * B.ctxU4 + 4];
                    const uint4* sl = B.A + CMB_IDX(sC4.x, B.aCap, 102);
                    for (uint32_t t = 0; t < P.nRem; t++) {
                        const uint32_t j = nd0 + t;
#pragma unroll
                        for (uint32_t u = 0; u < PU; u++) dl[DU * j + u] = sl[DU * ((uint32_t)remFrom + t) + u];
                        dl[DU * j + PU] = make_uint4((j + 1) | (sl[DU * ((uint32_t)remFrom + t) + PU].x & 0xFF0000u), 0u, 0u, 0u);
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
                uint16_t* il = reinterpret_cast<uint16_t*>(dl + DU * P.nDescNew);
                uint32_t mn = ieL[0][tid];
                for (uint32_t j = 0; j < nInitNew; j++) {
                    il[j] = ieL[j][tid];
                    mn = min(mn, (uint32_t)ieL[j][tid]);
                }
                if (s->dsw[idxN] && P.nDescNew > 0) { // :640-648
                    smR = Tr::load(dl + DU * (P.nDescNew - 1), 1);
                    smDist = mn;
                }
            }
        }

        // ---- phase entry: recApproxMatchEdit prologue + replay of the descendants (:377-497)
        uint32_t outKind = 0; // 1: node of the next frontier, 2: event
        uint32_t evRem = 0, evCell = 0, fLast = BFS_NONE; // event of an interrupted replay: descendants left, cell, its F record
        uint4 oN1 = make_uint4(0, 0, 0, 0), oN2 = oN1;
        EdPack oPack{}; // the final-column pack of the node / event this lane produces
        Pair oRoot = Tr::none();
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
            const uint4* dl = B.A + CMB_IDX(dListOff, B.aCap, 103);
            const uint16_t* il = reinterpret_cast<const uint16_t*>(dl + DU * nSrcDesc);
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
            initMatrix<Geo::LEFT, Geo::DIAG>(g, xLen, maxEDn, first, lastI, nSrcInit ? il : nullptr, increase, nInit, HP, HN, RAC, score);
            const uint32_t clSize = g.sfc();
            const uint32_t nBlk = (g.m - 1) / Geo::BLOCK + 1;
            if (g.Wv >= Geo::LEFT || clSize > ED_CELLS || nBlk > B.ctxMblk || g.m > 0xFFFFu) {
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
                            const uint4 oC4 = B.C[(size_t)CMB_IDX(otherRefN, B.cCap, 7) * B.ctxU4 + 4];
                            oOff = oC4.x;
                            oDesc = oC4.y & 0xFFu;
                            oInit = (oC4.y >> 8) & 0xFFu;
                        }
                        if (oDesc > 0) {
                            const uint16_t* oi = reinterpret_cast<const uint16_t*>(B.A + CMB_IDX(oOff + DU * oDesc, B.aCap, 104));
                            itStart -= oDesc - oInit + (uint32_t)oi[oInit - 1];
                        }
                    }
                    itMeta = packMeta(smShiftN, maxEDs, minEDs, stt == 0, ITEM_EDIT);
                }
                uint4* Cx = B.C + (size_t)CMB_IDX(cNew, B.cCap, 8) * B.ctxU4;
                Cx[0] = make_uint4(rsId, g.n | (g.m << 16), g.Wv | (g.Wh << 8) | (maxEDn << 16) | (clSize << 24),
                                   idxN | (dirN << 4) | (uniN << 5) | (useRev << 6) | (itMode << 7) | (scheme << 12) |
                                       (search << 16));
                Cx[1] = make_uint4(itStart, itMeta, descRefN, otherRefN);
                if (rsId > 0x1FFFFFFu || itMeta > 0x7FFFFFu) flags |= FLAG_CAPACITY;
                Cx[CTX_HOT] = make_uint4(rsId | (itMode << 25) | (dirN << 27) | (uniN << 28),
                                   g.n | (g.m << 9) | (g.Wv << 18) | (g.Wh << 23) | (maxEDn << 27), itStart,
                                   itMeta | (clSize << 23));
                Cx[3] = make_uint4(smDepthN, smShiftN, smDist, xOff | (xLen << 16));
                Cx[4] = make_uint4(descSelf || otherSelf ? aOff : 0u,
                                   descSelf || otherSelf ? (P.nDescNew | (nInitNew << 8)) : 0u, 0u, 0u);
                const uint32_t* Gs[4]; // the bit-strings of A, C, G, T for this read x strand and direction
                for (uint32_t c4 = 0; c4 < 4; c4++) Gs[c4] = gString(G, gw, rsId, (uint32_t)useRev, c4);
                for (uint32_t b = 0; b < nBlk; b++) {
                    const uint64_t a = matchWord<Geo::LEFT, Geo::BLOCK>(Gs[0], xOff, xLen, b), c = matchWord<Geo::LEFT, Geo::BLOCK>(Gs[1], xOff, xLen, b);
                    const uint64_t gg = matchWord<Geo::LEFT, Geo::BLOCK>(Gs[2], xOff, xLen, b), t = matchWord<Geo::LEFT, Geo::BLOCK>(Gs[3], xOff, xLen, b);
                    Cx[CTX_M + 2 * b] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)c, (uint32_t)(c >> 32));
                    Cx[CTX_M + 1 + 2 * b] = make_uint4((uint32_t)gg, (uint32_t)(gg >> 32), (uint32_t)t, (uint32_t)(t >> 32));
                }
                // first cell of the cluster (:452-461)
                EdPack pk{};
                uint32_t fcCur = BFS_NONE;
                if (g.inFinalColumn(0)) {
                    const uint32_t e0 = Geo::cell(0, xLen, HP, HN, score);
                    if (e0 > ED_MAX) flags |= FLAG_CAPACITY;
                    edPut(pk, 0, min(e0, ED_MAX));
                    uint4* Fr = B.F + (size_t)CMB_IDX(fNext, B.fCap, 21) * FU;
                    Tr::store(Fr, 1, smR);
                    Fr[PU] = make_uint4(0u, BFS_NONE, 0u, 0u);
                    fcCur = fNext++;
                }
                bool live = true;
                Pair root = smR;
                uint32_t rootRow = 0;
                if (nSrcDesc > 0) { // replay (:463-492)
                    const uint32_t maxRow = g.m - 1;
                    // (what the event of an interrupted replay carries is derived AFTER the loop from the loop counter and
                    // the allocation counter — values assigned inside this divergent loop and read long after it came
                    // back wrong for reads of 40 ... 100 characters: a record {0, 0, j + 1, cell})
                    uint32_t j = 0;
                    bool interrupted = false;
                    for (; j < nSrcDesc; j++) {
                        const uint32_t meta = dl[DU * j + PU].x;
                        const uint32_t depth = meta & 0xFFFFu, ch = (meta >> 16) & 0xFFu;
                        if (depth > maxRow) break;
                        const uint64_t M = matchWord<Geo::LEFT, Geo::BLOCK>(gString(G, gw, rsId, (uint32_t)useRev, ch - 1u), xOff, xLen, depth / Geo::BLOCK);
                        uint64_t D0;
                        const bool valid = Geo::row(g, depth, M, HP, HN, D0, RAC, score);
                        cRows++;
                        if (g.inFinalColumn(depth)) {
                            const uint32_t cellJ = clSize + depth - g.m;
                            const uint32_t e = Geo::cell(depth, g.n - 1, HP, HN, score);
                            if (e > ED_MAX) flags |= FLAG_CAPACITY;
                            edPut(pk, cellJ, min(e, ED_MAX));
                            uint4* Fr = B.F + (size_t)CMB_IDX(fNext, B.fCap, 22) * FU;
#pragma unroll
                            for (uint32_t u = 0; u < PU; u++) Fr[u] = dl[DU * j + u];
                            Fr[PU] = make_uint4(depth | (ch << 16), fcCur, 0u, 0u);
                            fcCur = fNext++;
                            if (!valid || Geo::ovgl(g, depth, HN)) { // goDeeper, then `return` (:472-477)
                                interrupted = true;
                                live = false;
                                break;
                            }
                        }
                        if (!valid) {
                            live = false;
                            break;
                        }
                    }
                    if (interrupted) {
                        outKind = 2;
                        evRem = j + 1;
                        evCell = clSize + (dl[DU * j + PU].x & 0xFFFFu) - g.m;
                        fLast = fNext - 1u; // (the F record of that row was the last one handed out)
                        oPack = pk;
                    }
                    if (live) {
                        const uint32_t lastDepth = dl[DU * (nSrcDesc - 1) + PU].x & 0xFFFFu;
                        if (lastDepth == maxRow) live = false; // :479
                        else {
                            rootRow = lastDepth;
                            if (!dswN) { // after a switch the range of the start match is kept (:485)
                                root = Tr::load(dl + DU * (nSrcDesc - 1), 1);
                            }
                        }
                    }
                }
                if (live) {
                    outKind = 1;
                    oRoot = root;
                    oN1 = make_uint4(rootRow | (score << 16), cNew, fcCur,
                                     ((uint32_t)__ffsll((unsigned long long)RAC) - 1u) | ((uniN ? 2u : (dirN == 0 ? 0u : 1u)) << 8));
                    oN2 = make_uint4((uint32_t)HP, (uint32_t)(HP >> 32), (uint32_t)HN, (uint32_t)(HN >> 32));
                    oPack = pk;
                }
            }
        }
        // ---- append the node / event this lane produced
        uint32_t t4, t5;
        const uint32_t oN = Tr::CHUNKED_Q ? blockAppendChunked(&B.nq[outP], outKind == 1 ? 1u : 0u, sh[0], t4, B.qCnt[outP & 1u], qCap)
                                          : blockAppend(&B.nq[outP], outKind == 1 ? 1u : 0u, sh[0], t4);
        const uint32_t oE = blockAppend(&B.ne[outP], outKind == 2 ? 1u : 0u, sh[1], t5);
        if (outKind == 1) {
            if (oN >= qCap) flags |= FLAG_BFS_Q;
            else {
                Tr::store(Qo + oN, qCap, oRoot);
                Qo[(size_t)PU * qCap + oN] = oN1;
                Qo[(size_t)(PU + 1) * qCap + oN] = oN2;
                packStore(Qo + (size_t)(PU + 2) * qCap + oN, qCap, oPack);
            }
        } else if (outKind == 2) {
            if (oE >= B.evCap) flags |= FLAG_BFS_EV;
            else {
                Eo[(size_t)EV_U4 * oE] = make_uint4(cNew, fLast, evRem, evCell);
                packStore(Eo + (size_t)EV_U4 * oE + 1, 1, oPack);
            }
        }
    }
    unsigned long long v = cRows;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((tid & 63u) == 0 && v) atomicAdd(&q.counters[11], v);
    if (flags) atomicOr(&q.cnt[3], flags);
}

} // namespace cmb

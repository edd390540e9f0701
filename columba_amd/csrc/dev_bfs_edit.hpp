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
constexpr uint32_t BFS_CHAIN = 5;     // expansions a lane makes in a row while each yields exactly one plain node (4 before the chain
                                      // could cross the 8-row blocks of the small matrix: 57.2 -> 55.8 ms)
constexpr uint32_t BFS_GRID = 1024;   // blocks of the grid-stride frontier kernels without an event half (k_hbfs, k_naive_pass, ...)
constexpr uint32_t BFS_GRID_CNT = 2048; // MOST blocks that expand the frontier in k_bfs_pass (CMB_BFS_GRID); sizes the per-block counters
constexpr uint32_t BFS_GRID_X = 896;  // default: expanding blocks (k_bfs_pass runs four 256-thread blocks per CU) ...
constexpr uint32_t BFS_GRID_EV = 128; // ... and blocks that handle the events of the same pass

// (the FLAG_BFS_* bits live in dev_search.hpp with all other bits of the flag word)
// any of these set by an earlier pass: the frontier is incomplete, later passes do nothing (the host re-runs)
constexpr uint32_t BFS_STOP = FLAG_BFS_Q | FLAG_BFS_EV | FLAG_BFS_F | FLAG_BFS_CTX | FLAG_BFS_ARENA | FLAG_ITEM_OVERFLOW |
                              FLAG_FMOCC_OVERFLOW | FLAG_CAPACITY | FLAG_NARROW_MATRIX;

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
    unsigned long long* blockCnt; // [BFS_GRID_CNT][4] per-block counters: nodes, expansions, rows, -
    uint32_t narrowWv;            // GeoN32: widest first column (Wv) a phase may have on the small matrix (its DIAG; tests lower it)
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
struct FmTraits {
    typedef RangePair Pair;
    typedef DfsTask Task;
    static constexpr uint32_t PAIR_U4 = 1;
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
// What a geometry says about its matrix: the word type W of a row's state (HP, HN; the rightmost active column is a one-bit mask of that
// type, kept in records as its bit index), rows per block, left margin and diagonal offset, and CTX_BLOCK / CTX_LEFT: the blocks and the
// margin of the 64-bit MATCH WORDS its contexts hold (`mword`: the match word of a row from its context block's word).
struct MxRef64 {
    typedef uint64_t W;
    static constexpr uint32_t BLOCK = MX_BLOCK, LEFT = MX_LEFT, DIAG = MX_DIAG, CTX_BLOCK = MX_BLOCK, CTX_LEFT = MX_LEFT;
    static constexpr bool NARROW_FALLBACK = false;
    static __device__ __forceinline__ W mword(uint64_t M, uint32_t) { return M; }
    static __device__ __forceinline__ W racBit(uint32_t idx) { return 1ull << idx; }
    static __device__ __forceinline__ uint32_t racIdx(W rac) { return (uint32_t)__ffsll((unsigned long long)rac) - 1u; }
    static __device__ __forceinline__ uint4 packRow(W HP, W HN) { return make_uint4((uint32_t)HP, (uint32_t)(HP >> 32), (uint32_t)HN, (uint32_t)(HN >> 32)); }
    static __device__ __forceinline__ void unpackRow(const uint4& v, W& HP, W& HN) {
        HP = (uint64_t)v.x | ((uint64_t)v.y << 32);
        HN = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
    static __device__ __forceinline__ bool row(const MatGeom& g, uint32_t i, uint64_t M, uint64_t& HP, uint64_t& HN, uint64_t& D0, uint64_t& RAC,
                                               uint32_t& sc) {
        return computeRow(g, i, M, HP, HN, D0, RAC, sc);
    }
    static __device__ __forceinline__ uint32_t cell(uint32_t i, uint32_t j, uint64_t HP, uint64_t HN, uint32_t sc) { return cellAt(i, j, HP, HN, sc); }
    static __device__ __forceinline__ bool ovgl(const MatGeom& g, uint32_t i, uint64_t HN) { return onlyVerticalGapsLeft(g, i, HN); }
};
struct MxNarrow {
    typedef uint64_t W;
    static constexpr uint32_t BLOCK = MXN_BLOCK, LEFT = MXN_LEFT, DIAG = MXN_DIAG, CTX_BLOCK = MXN_BLOCK, CTX_LEFT = MXN_LEFT;
    static constexpr bool NARROW_FALLBACK = false;
    static __device__ __forceinline__ W mword(uint64_t M, uint32_t) { return M; }
    static __device__ __forceinline__ W racBit(uint32_t idx) { return 1ull << idx; }
    static __device__ __forceinline__ uint32_t racIdx(W rac) { return (uint32_t)__ffsll((unsigned long long)rac) - 1u; }
    static __device__ __forceinline__ uint4 packRow(W HP, W HN) { return make_uint4((uint32_t)HP, (uint32_t)(HP >> 32), (uint32_t)HN, (uint32_t)(HN >> 32)); }
    static __device__ __forceinline__ void unpackRow(const uint4& v, W& HP, W& HN) {
        HP = (uint64_t)v.x | ((uint64_t)v.y << 32);
        HN = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
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
// the in-index matrix up to 6 errors on 32-bit words and 8-row blocks (dev_matrix.hpp: MXS_*): half the instructions of a matrix row; the
// contexts keep the 64-bit match words of 32-row blocks, a lane keeps walking its chain across the 8-row blocks inside them
struct MxSmall32 {
    typedef uint32_t W;
    static constexpr uint32_t BLOCK = MXS_BLOCK, LEFT = MXS_LEFT, DIAG = MXS_DIAG, CTX_BLOCK = MX_BLOCK, CTX_LEFT = MX_LEFT;
    static constexpr bool NARROW_FALLBACK = true; // (a phase with Wv > DIAG - MXS_SLACK: FLAG_NARROW_MATRIX, the host re-runs on GeoN)
    static __device__ __forceinline__ W mword(uint64_t M, uint32_t i) { return matchWordSmall(M, i); }
    static __device__ __forceinline__ W racBit(uint32_t idx) { return 1u << idx; }
    static __device__ __forceinline__ uint32_t racIdx(W rac) { return 31u - (uint32_t)__clz((int)rac); }
    static __device__ __forceinline__ uint4 packRow(W HP, W HN) { return make_uint4(HP, HN, 0u, 0u); }
    static __device__ __forceinline__ void unpackRow(const uint4& v, W& HP, W& HN) {
        HP = v.x;
        HN = v.y;
    }
    static __device__ __forceinline__ bool row(const MatGeom& g, uint32_t i, W M, W& HP, W& HN, W& D0, W& RAC, uint32_t& sc) {
        return computeRow<BLOCK, DIAG>(g, i, M, HP, HN, D0, RAC, sc);
    }
    static __device__ __forceinline__ uint32_t cell(uint32_t i, uint32_t j, W HP, W HN, uint32_t sc) { return cellAt<BLOCK, DIAG>(i, j, HP, HN, sc); }
    static __device__ __forceinline__ bool ovgl(const MatGeom& g, uint32_t i, W HN) { return onlyVerticalGapsLeftAs<BLOCK, DIAG>(g, i, HN, 64u); }
};
struct GeoN : MxRef64 {
    static constexpr int MP = MAXP;
    typedef EdPack Pack;
    static constexpr uint32_t CELLS = 24, PK_U4 = 1, ED_MAX = 31;
};
struct GeoN32 : MxSmall32 { // the common path since round 4: GeoN's records (the row state takes two words of its plane), the small matrix
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
// wavefronts per SIMD = 99 / 71 ms per step; a fifth wavefront — 96 registers, round 4 — bought nothing more).  Hence a lane
// never HOLDS the four children: it classifies them (evalChildFlat: 4 bits per child, keeping the state of the last one that
// leaves a record — THE child of a lane that walks on), the block allocates the queue slots, and the children that leave are
// computed again straight into their records.
enum : uint32_t { KIND_NONE = 0, KIND_NODE = 1, KIND_EVENT = 2, KIND_ITEMS = 3 };
struct ExpandCtx { // what the expansions of one phase share (from the context's hot word)
    MatGeom g;
    uint32_t clSize, itMode, itStart, switchPoint;
};
template <typename W = uint64_t>
struct ChildStateT {
    RangePair r;
    W HP, HN, RAC;
    uint32_t sc, aux; // aux: final-column distance of the child (needF) or in-text start difference (KIND_ITEMS)
};
typedef ChildStateT<> ChildState;
// Kind of child `ch` of `parent` at row `row1` | needF << 2 | capacity problem << 3, and its state: the child's matrix row (computeRow),
// branchAndBound (:529-561), the in-text switch (:340-375, :516).  Written WITHOUT nested control flow (round 4): as a nest of conditions —
// final column or not, valid or not, gaps, in-text switch, extension mode — a child cost a wavefront ~205 instructions of which ~115
// were the exec-mask bookkeeping of the nest.  Here the final-column distance, the gaps predicate and the in-text start difference are
// computed for every child (a few vector instructions each on the small matrix) and the outcome is selected; what is left of the control
// flow is "does the child exist" and the walk to the rightmost active column of a row whose RAC column missed: ~165 instructions per
// child, 3 200 instead of 3 800 per tile.  (The time did not follow: see profiles/r04_frontier_experiments.txt.)  `out` is written only
// for a child that leaves a record (kind != KIND_NONE); `width` is the child's range width (in-text items).
template <class Geo = GeoN>
__device__ __forceinline__ uint32_t evalChildFlat(const DevIndex& ix, int md, const RangePair& parent, uint32_t ch, const uint32_t Rb[4],
                                                  const uint32_t Re[4], uint32_t db, uint32_t de, const ExpandCtx& e, uint32_t row1, bool inFC,
                                                  typename Geo::W M, typename Geo::W pHP, typename Geo::W pHN, typename Geo::W pRAC,
                                                  uint32_t score, bool& nonEmpty, ChildStateT<typename Geo::W>& out, uint32_t& width) {
    typedef typename Geo::W W;
    RangePair child;
    nonEmpty = childFromRanksFlat(ix, md, parent, ch, Rb, Re, db, de, child);
    width = 0;
    if (!nonEmpty) return KIND_NONE;
    W HP = pHP, HN = pHN, RAC = pRAC, D0;
    uint32_t sc = score;
    constexpr uint32_t EDMAX = Geo::ED_MAX;
    const bool valid = Geo::row(e.g, row1, M, HP, HN, D0, RAC, sc);
    const uint32_t ed = Geo::cell(row1, e.g.n - 1, HP, HN, sc); // (meaningful in the final column only)
    const bool gaps = Geo::ovgl(e.g, row1, HN);
    const bool live = valid || inFC;                            // else pruned when popped (branchAndBound returns true, :560)
    const bool event = inFC && (!valid || gaps);                // goDeeper
    width = child.sa.e - child.sa.b;
    const bool items = live && !event && width <= e.switchPoint && e.itMode != 0; // goToInTextVerificationEdit (:340-375)
    const uint32_t col = e.g.firstColumn(row1);
    const uint32_t startDiff = e.itStart - (e.itMode == 2 ? col + Geo::cell(row1, col, HP, HN, sc) : 0u);
    uint32_t res = items ? (uint32_t)KIND_ITEMS : event ? (uint32_t)KIND_EVENT : (uint32_t)KIND_NODE;
    res |= (inFC && !items) ? 4u : 0u;    // (a child that leaves the index needs no F record)
    res |= (inFC && ed > EDMAX) ? 8u : 0u;
    res = live ? res : (uint32_t)KIND_NONE;
    if (live) {
        out.r = child;
        out.HP = HP;
        out.HN = HN;
        out.RAC = RAC;
        out.sc = sc;
        out.aux = items ? startDiff : inFC ? min(ed, EDMAX) : 0u;
    }
    return res;
}

template <class Geo = GeoN>
__device__ __forceinline__ void bfsExpand(const DevIndex& ix, const BfsBufs& B, uint32_t pass, const Queues& q,
                                          uint32_t bid, uint32_t nBlocks) {
    typedef typename Geo::W W;
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
        W pHP = 0, pHN = 0;
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
            blk = (row + 1) / Geo::CTX_BLOCK;
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
            Geo::unpackRow(n2, pHP, pHN);
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
                ChildStateT<W> one{}; // the last child that leaves a record: THE child when the expansion yields exactly one plain node
#pragma unroll
                for (uint32_t ch = 1; ch <= 4; ch++) {
                    bool nonEmpty;
                    uint32_t width;
                    const uint32_t k4 = evalChildFlat<Geo>(ix, md, parent, ch, Rb, Re, db, de, e, row1, inFC, Geo::mword(ldsM[ch - 1][tid], row1), pHP,
                                                           pHN, Geo::racBit(pRac), score, nonEmpty, one, width);
                    nChildren += nonEmpty ? 1u : 0u;
                    flags |= (k4 & 8u) ? (uint32_t)FLAG_CAPACITY : 0u;
                    kinds |= (k4 & 7u) << (4 * (ch - 1));
                    nOut += (k4 & 3u) != KIND_NONE ? 1u : 0u;
                    nIt += (k4 & 3u) == KIND_ITEMS ? width : 0u;
                }
                ldsCnt[0][tid] += nChildren;
                // exactly one child, a plain node, with rows left in this block of match words: keep walking
                const bool single = nOut == 1u && (kinds == 0x1u || kinds == 0x10u || kinds == 0x100u || kinds == 0x1000u);
                if (single && step + 1u < B.chain && (row1 + 1u) / Geo::CTX_BLOCK == blk) {
                    parent = one.r;
                    score = one.sc;
                    pHP = one.HP;
                    pHN = one.HN;
                    pRac = Geo::racIdx(one.RAC);
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
                uint32_t width;
                ChildStateT<W> cs{};
                (void)evalChildFlat<Geo>(ix, md, parent, ch, Rb, Re, db, de, e, row1, inFC, Geo::mword(ldsM[ch - 1][tid], row1), pHP, pHN,
                                         Geo::racBit(pRac), score, nonEmpty, cs, width);
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
                    Qo[(size_t)qCap + o] = make_uint4(row1 | (cs.sc << 16), ctx, fc, Geo::racIdx(cs.RAC) | ((uint32_t)md << 8));
                    Qo[(size_t)2 * qCap + o] = Geo::packRow(cs.HP, cs.HN);
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
            } else if (Ei[(size_t)EV_U4 * i].x != BFS_NONE) { // (a record without a context is a hole)
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
            typename Geo::W HP, HN, RAC;
            uint32_t score;
            initMatrix<Geo::LEFT, Geo::DIAG, typename Geo::W>(g, xLen, maxEDn, first, lastI, nSrcInit ? il : nullptr, increase, nInit, HP, HN, RAC, score);
            const uint32_t clSize = g.sfc();
            const uint32_t nBlk = (g.m - 1) / Geo::CTX_BLOCK + 1;
            if (Geo::NARROW_FALLBACK && (g.Wv > min(Geo::DIAG - MXS_SLACK, B.narrowWv) || maxEDn > MXS_MAX_ED)) {
                flags |= FLAG_NARROW_MATRIX; // (the first column does not fit the small matrix: the host runs the batch on the 64-bit geometry)
            } else if (g.Wv >= Geo::LEFT || clSize > ED_CELLS || nBlk > B.ctxMblk || g.m > 0xFFFFu) {
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
                    const uint64_t a = matchWord<Geo::CTX_LEFT, Geo::CTX_BLOCK>(Gs[0], xOff, xLen, b), c = matchWord<Geo::CTX_LEFT, Geo::CTX_BLOCK>(Gs[1], xOff, xLen, b);
                    const uint64_t gg = matchWord<Geo::CTX_LEFT, Geo::CTX_BLOCK>(Gs[2], xOff, xLen, b), t = matchWord<Geo::CTX_LEFT, Geo::CTX_BLOCK>(Gs[3], xOff, xLen, b);
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
                        const typename Geo::W M = Geo::mword(
                            matchWord<Geo::CTX_LEFT, Geo::CTX_BLOCK>(gString(G, gw, rsId, (uint32_t)useRev, ch - 1u), xOff, xLen, depth / Geo::CTX_BLOCK), depth);
                        typename Geo::W D0;
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
                    oN1 = make_uint4(rootRow | (score << 16), cNew, fcCur, Geo::racIdx(RAC) | ((uniN ? 2u : (dirN == 0 ? 0u : 1u)) << 8));
                    oN2 = Geo::packRow(HP, HN);
                    oPack = pk;
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

// The approximate search over the run-length compressed backend (b-move; SURVEY.md §8 row f3, BASELINE configs[4]).
//
// What the reference compiles with -DRUN_LENGTH_COMPRESSION is the SAME search layer over another index
// (IndexInterface / SearchStrategy with BMove behind the virtuals, paths relative to /root/reference/src):
//   ranges carry run indices, a toehold, toeholdRepresentsEnd and originalDepth      indexhelpers.h:137-255, :1040-1260
//   no in-text verification: switch point 0, goToInTextVerificationEdit returns      indexinterface.cpp:345-348, :516-524,
//     at once, no part-level pre-verification, every search start is SEARCH_STARTED    :1306-1325; searchstrategy.cpp:461-477;
//                                                                                      bmove.cpp:195-197
//   k-mer table entries with exact run indices of the SA range                        indexinterface.cpp:327-329
//   in-index occurrences -> text positions by the toehold's phi / phi^-1 chains       bmove.cpp:500-560
// Here that becomes: the frontier of dev_bfs_edit.hpp (events, phase entry, cluster analysis: bfsHeavy, instantiated with
// MvTraits — nodes, F records and descendant lists hold move range pairs packed into 48 bytes instead of four 32-bit bounds), an
// expansion step built on moveChildren (all four children of a node from one scan of the parent's runs), a prologue
// (partitioning, exact phases) on the same primitive, and a post-processing chain: de-duplication of the in-index
// occurrences, k_move_locate, sort + redundancy filter on 64-bit positions.
#pragma once
#include <type_traits>
#include "dev_bfs_edit.hpp"
#include "move_dev.hpp"

namespace cmb {

// Frontier records are written once and read once, a pass later: with CMB_MVS_NT (A/B build) their loads and stores carry the
// non-temporal hint, so that they do not push move-table rows out of the L2 and the Infinity Cache.
typedef uint32_t mvs_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 qLoad(const uint4* p) {
#ifdef CMB_MVS_NT
    const mvs_v4 v = __builtin_nontemporal_load(reinterpret_cast<const mvs_v4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void qStore(uint4* p, const uint4& a) {
#ifdef CMB_MVS_NT
    mvs_v4 v;
    v.x = a.x, v.y = a.y, v.z = a.z, v.w = a.w;
    __builtin_nontemporal_store(v, reinterpret_cast<mvs_v4*>(p));
#else
    *p = a;
#endif
}

struct MvTask { // a search that starts its first approximate phase (k_mvs_exact -> k_mvs_start)
    uint32_t rsId;
    uint8_t scheme, search, idx, pad;
    uint32_t depth, pad2;
    MoveRangeRec r;
};
static_assert(sizeof(MvTask) == 96, "record layout");

struct MvFmRec { // in-index occurrence (FMOcc of the RLC flavour: the ranges travel with toehold and original depth)
    uint32_t rsId, depth, dist, shift;
    MoveRangeRec r;
};
static_assert(sizeof(MvFmRec) == 96, "record layout");

struct MvBufs : BfsBufs {
    MvFmRec* fmX;                  // in-index occurrences (slots handed out through Queues::cnt[1], capacity Queues::fmCap)
    unsigned long long* rowSteps;  // [BFS_GRID] table rows fetched by the expansions (this backend's byte-model unit)
};

// In the records of the frontier (node planes, F records, descendant lists) a range pair is packed into THREE uint4: nine 40-bit
// fields (both ranges with their run indices, the toehold: texts and tables below 2^40, checked at index creation), the 16-bit
// original depth and the three flags — 379 of 384 bits, against the 80-byte cmb_move_range of the C-ABI.
struct MvTraits {
    typedef MvPair Pair;
    typedef MvTask Task;
    static constexpr uint32_t PAIR_U4 = 3;
    static __device__ __forceinline__ Pair unpack(const uint4& a, const uint4& b, const uint4& c) {
        const uint64_t w0 = u64of(a.x, a.y), w1 = u64of(a.z, a.w), w2 = u64of(b.x, b.y), w3 = u64of(b.z, b.w), w4 = u64of(c.x, c.y),
                       w5 = u64of(c.z, c.w);
        Pair r;
        r.sa.begin = w0 & MV_M40;
        r.sa.end = (w0 >> 40 | w1 << 24) & MV_M40;
        r.sa.beginRun = (w1 >> 16) & MV_M40;
        r.sa.endRun = (w1 >> 56 | w2 << 8) & MV_M40;
        r.rev.begin = (w2 >> 32 | w3 << 32) & MV_M40;
        r.rev.end = (w3 >> 8) & MV_M40;
        r.rev.beginRun = (w3 >> 48 | w4 << 16) & MV_M40;
        r.rev.endRun = (w4 >> 24) & MV_M40;
        r.toehold = w5 & MV_M40;
        r.depth = (uint32_t)(w5 >> 40) & 0xFFFFu;
        r.sa.valid = (w5 >> 56) & 1u;
        r.rev.valid = (w5 >> 57) & 1u;
        r.repEnd = (w5 >> 58) & 1u;
        return r;
    }
    static __device__ __forceinline__ void pack(const Pair& r, uint4& a, uint4& b, uint4& c) {
        const uint64_t w0 = (r.sa.begin & MV_M40) | r.sa.end << 40;
        const uint64_t w1 = (r.sa.end & MV_M40) >> 24 | (r.sa.beginRun & MV_M40) << 16 | r.sa.endRun << 56;
        const uint64_t w2 = (r.sa.endRun & MV_M40) >> 8 | r.rev.begin << 32;
        const uint64_t w3 = (r.rev.begin & MV_M40) >> 32 | (r.rev.end & MV_M40) << 8 | r.rev.beginRun << 48;
        const uint64_t w4 = (r.rev.beginRun & MV_M40) >> 16 | (r.rev.endRun & MV_M40) << 24;
        const uint64_t w5 = (r.toehold & MV_M40) | (uint64_t)(r.depth & 0xFFFFu) << 40 | (uint64_t)(r.sa.valid ? 1 : 0) << 56 |
                            (uint64_t)(r.rev.valid ? 1 : 0) << 57 | (uint64_t)(r.repEnd ? 1 : 0) << 58;
        a = make_uint4((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32));
        b = make_uint4((uint32_t)w2, (uint32_t)(w2 >> 32), (uint32_t)w3, (uint32_t)(w3 >> 32));
        c = make_uint4((uint32_t)w4, (uint32_t)(w4 >> 32), (uint32_t)w5, (uint32_t)(w5 >> 32));
    }
    // the same records from / into 32-bit positions (indexes below 2^32: the nine 40-bit fields sit at bits 0, 40, ..., 320 — twelve dwords,
    // every field the funnel shift of two of them; the high byte of a field is zero)
    static __device__ __forceinline__ MvPairT<uint32_t> unpack32(const uint4& a, const uint4& b, const uint4& c) {
        MvPairT<uint32_t> r;
        r.sa.begin = a.x;
        r.sa.end = __funnelshift_r(a.y, a.z, 8u);
        r.sa.beginRun = __funnelshift_r(a.z, a.w, 16u);
        r.sa.endRun = __funnelshift_r(a.w, b.x, 24u);
        r.rev.begin = b.y;
        r.rev.end = __funnelshift_r(b.z, b.w, 8u);
        r.rev.beginRun = __funnelshift_r(b.w, c.x, 16u);
        r.rev.endRun = __funnelshift_r(c.x, c.y, 24u);
        r.toehold = c.z;
        r.depth = (c.w >> 8) & 0xFFFFu;
        r.sa.valid = (c.w >> 24) & 1u;
        r.rev.valid = (c.w >> 25) & 1u;
        r.repEnd = (c.w >> 26) & 1u;
        return r;
    }
    static __device__ __forceinline__ void pack(const MvPairT<uint32_t>& r, uint4& a, uint4& b, uint4& c) {
        a = make_uint4(r.sa.begin, r.sa.end << 8, (r.sa.end >> 24) | (r.sa.beginRun << 16), (r.sa.beginRun >> 16) | (r.sa.endRun << 24));
        b = make_uint4(r.sa.endRun >> 8, r.rev.begin, r.rev.end << 8, (r.rev.end >> 24) | (r.rev.beginRun << 16));
        c = make_uint4((r.rev.beginRun >> 16) | (r.rev.endRun << 24), r.rev.endRun >> 8, r.toehold,
                       ((r.depth & 0xFFFFu) << 8) | ((r.sa.valid ? 1u : 0u) << 24) | ((r.rev.valid ? 1u : 0u) << 25) | ((r.repEnd ? 1u : 0u) << 26));
    }
    template <typename P> static __device__ __forceinline__ MvPairT<P> unpackT(const uint4& a, const uint4& b, const uint4& c);
    static __device__ __forceinline__ Pair load(const uint4* p, size_t stride) { return unpack(p[0], p[stride], p[2 * stride]); }
    static __device__ __forceinline__ void store(uint4* p, size_t stride, const Pair& r) {
        uint4 a, b, c;
        pack(r, a, b, c);
        p[0] = a;
        p[stride] = b;
        p[2 * stride] = c;
    }
    static __device__ __forceinline__ Pair none() {
        Pair p;
        p.sa = {0, 0, 0, 0, true};
        p.rev = p.sa;
        p.toehold = 0, p.repEnd = false, p.depth = 0;
        return p;
    }
    static __device__ __forceinline__ bool empty(const Pair& r) { return r.sa.end <= r.sa.begin; }
    static __device__ __forceinline__ Pair taskRange(const Task& t) { return loadPair(t.r); }
    template <class BUFS>
    static __device__ __forceinline__ void emitFm(const BUFS& B, const Queues&, uint32_t slot, uint32_t rsId, const Pair& r, uint32_t depth,
                                                  uint32_t ed, uint32_t shift) {
        MvFmRec f;
        f.rsId = rsId, f.depth = depth, f.dist = ed, f.shift = shift;
        f.r = storePair(r);
        static_cast<const MvBufs&>(B).fmX[slot] = f;
    }
    template <class BUFS> static __device__ __forceinline__ void fmHole(const BUFS& B, const Queues&, uint32_t slot) {
        static_cast<const MvBufs&>(B).fmX[slot].rsId = 0xFFFFFFFFu;
    }
};

template <> __device__ __forceinline__ MvPairT<uint64_t> MvTraits::unpackT<uint64_t>(const uint4& a, const uint4& b, const uint4& c) { return unpack(a, b, c); }
template <> __device__ __forceinline__ MvPairT<uint32_t> MvTraits::unpackT<uint32_t>(const uint4& a, const uint4& b, const uint4& c) { return unpack32(a, b, c); }

// ---- the index as the search sees it
struct MvSearchIndex {
    MoveDev d;
    const MoveRangeRec* kmer; // 4^kmerSize entries (k_move_kmer_table)
    uint32_t kmerSize;
};

// row fetches of the walks, for the byte model (counted where the rows are loaded)
struct RowCount {
    uint32_t n = 0;
};

// moveChildren (move_dev.hpp) arranged for the memory system, with a count of the table rows it fetches:
//  * the two ends of the range are scanned TOGETHER (one loop, two cursors: their row loads overlap);
//  * a row that holds the first / last occurrence of a character is mapped by LF when it is seen (the row is in registers:
//    no second fetch), so the scan leaves, per character, the LF images of its first and last occurrence and their runs;
//    countChar (moverepr.cpp:329-345) = the distance of the two images — the widths of ALL children and the cumulative
//    counts cost no further memory access;
//  * only the fast-forwards (moverepr.cpp:283-293) of the children in `need` are walked, all of them in one loop (every
//    round issues the next row of every unfinished end point before any reply is consumed).
// Everything indexed by character is unrolled: the state stays in registers.  `need`: bit c - 1 = child c is wanted in
// full; the others only contribute their width to the cumulative counts.  Returns the mask of non-empty children.
// (a range chosen field by field: `cond ? a : b` on the objects makes the compiler choose between two ADDRESSES, which puts both
// objects — and whatever they were copied from — into scratch memory)
template <typename P>
__device__ __forceinline__ MvRangeT<P> selRange(bool c, const MvRangeT<P>& a, const MvRangeT<P>& b) {
    MvRangeT<P> r;
    r.begin = c ? a.begin : b.begin;
    r.end = c ? a.end : b.end;
    r.beginRun = c ? a.beginRun : b.beginRun;
    r.endRun = c ? a.endRun : b.endRun;
    r.valid = c ? a.valid : b.valid;
    return r;
}
// PACK: the children go straight into their packed form (MvTraits::pack, three uint4 each: twelve registers instead of twenty-one), one
// after the other — the four unpacked pairs never exist side by side (`child` is not touched).
// P: the type of positions and run numbers (uint64_t; uint32_t for the frontier kernel on indexes below 2^32 — half the registers of the
// scan's state and 32-bit instead of 64-bit arithmetic throughout).
template <bool PACK = false, typename P = uint64_t>
__device__ __forceinline__ uint32_t moveChildrenCounted(const MoveDev& ix, const int mode, const MvPairT<P>& parent, MvPairT<P> child[4], uint32_t& rows,
                                               const uint32_t need = 0xFu, uint4 (*pk)[3] = nullptr) {
    const bool fw = mode == 0;
    // the table of the direction, chosen FIELD BY FIELD: a reference `fw ? ix.rev : ix.fwd` is a choice between two addresses — every use of
    // a field then fetched the field from memory first (a pointer load and a wait in front of every sample and row access)
    MoveTable t;
    t.rows = fw ? ix.rev.rows : ix.fwd.rows;
    t.runs = fw ? ix.rev.runs : ix.fwd.runs;
    t.zeroCharPos = fw ? ix.rev.zeroCharPos : ix.fwd.zeroCharPos;
    t.samplesFirst = fw ? ix.rev.samplesFirst : ix.fwd.samplesFirst;
    t.samplesLast = fw ? ix.rev.samplesLast : ix.fwd.samplesLast;
    MvRangeT<P> trivial = selRange(fw, parent.rev, parent.sa);
    const MvRangeT<P> other = selRange(fw, parent.sa, parent.rev);
    if (!trivial.valid) { // (two binary searches between the enclosing run indices)
        P span = trivial.endRun - trivial.beginRun;
        while (span) {
            rows += 2;
            span >>= 1;
        }
        computeRunIndices(t, trivial);
    }
    P fOut[4], fRun[4], lOut[4], lRun[4], lSrc[4]; // LF images of the first / last occurrence, run of the last occurrence
    uint32_t found = 0, seen = 0;                          // bits 0..4: character seen from the front / from the back
    uint32_t ffNeed = 0;                                   // end points (2 c: first, 2 c + 1: last) whose image left its target run
    {
        P runF = trivial.beginRun, posF = trivial.begin, runB = trivial.endRun, posB = trivial.end - 1;
        uint4 rowF = t.rows[runF], rowB = t.rows[runB];
        // the rows the two cursors move to next are requested WITH the first two (a range within one run asks for its own row again: the
        // same address): a scan of up to four rows — the average is 3.6 — costs one memory round trip, and every later turn finds its rows
        // requested a turn earlier.  Only rows a cursor moves to are counted.
        uint4 preF = t.rows[runF < trivial.endRun ? runF + 1 : runF], preB = t.rows[runB > trivial.beginRun ? runB - 1 : runB];
        rows += 2;
        bool fDone = false, bDone = false;
        while (true) {
            if (!fDone) {
                const uint32_t h = rowHead(rowF);
                if (!(found >> h & 1u)) {
                    found |= 1u << h;
                    const MoveRowT<P> r = unpackMoveRowT<P>(rowF);
#pragma unroll
                    for (uint32_t c = 0; c < 4; c++)
                        if (h == c + 1) {
                            const P off = posF - r.in;
                            const uint32_t gap = rowGap(rowF);
                            fOut[c] = r.out + off;
                            fRun[c] = r.outRun;
                            if (off >= gap) { // (the row's gap field, move_dev.hpp: most images stay in the target run)
                                ffNeed |= 1u << (2 * c);
                                if (gap < MV_GAP_MAX) fRun[c]++;
                            }
                        }
                }
                fDone = (found & 0x1Eu) == 0x1Eu || runF == trivial.endRun;
            }
            if (!bDone) {
                const uint32_t h = rowHead(rowB);
                if (!(seen >> h & 1u)) {
                    seen |= 1u << h;
                    const MoveRowT<P> r = unpackMoveRowT<P>(rowB);
#pragma unroll
                    for (uint32_t c = 0; c < 4; c++)
                        if (h == c + 1) {
                            const P off = posB - r.in;
                            const uint32_t gap = rowGap(rowB);
                            lOut[c] = r.out + off;
                            lRun[c] = r.outRun;
                            lSrc[c] = runB;
                            if (off >= gap) {
                                ffNeed |= 2u << (2 * c);
                                if (gap < MV_GAP_MAX) lRun[c]++;
                            }
                        }
                }
                // every character of the forward pass is met again at the latest when the walk reaches the run that pass
                // stopped in; characters it did not meet (the forward pass ended early with all four) cannot exist
                bDone = (seen & 0x1Eu) == 0x1Eu || runB == trivial.beginRun || (fDone && (seen & 0x1Eu) == (found & 0x1Eu));
            }
            if (fDone && bDone) break;
            // the next rows of both cursors are REQUESTED TOGETHER: nothing between the two loads reads a reply (with `posF = rowIn(rowF)`
            // between them hipcc waited for the front row before it asked for the back row: two round trips per step of the scan)
            const P posBNext = rowInT<P>(rowB) - 1;
            if (!fDone) {
                runF++;
                rowF = preF;
                rows++;
                posF = rowInT<P>(rowF);
            }
            if (!bDone) {
                posB = posBNext;
                runB--;
                rowB = preB;
                rows++;
            }
            // (unconditional loads, clamped to the range's runs: a cursor that has finished asks for its last row again)
            preF = t.rows[runF < trivial.endRun ? runF + 1 : runF];
            preB = t.rows[runB > trivial.beginRun ? runB - 1 : runB];
        }
        found &= seen | 1u;
    }
    // fast-forwards of the wanted children: end point 2 c = first occurrence of character c + 1, 2 c + 1 = last
    {
        uint32_t act = 0;
#pragma unroll
        for (uint32_t c = 0; c < 4; c++)
            if ((found >> (c + 1) & 1u) && (need >> c & 1u)) act |= 3u << (2 * c);
        act &= ffNeed;
        while (act) {
            // (the raw words are requested first, all of them, and looked at afterwards: `rowIn(t.rows[...])` inside the condition made
            // hipcc wait for every one of the up to eight loads before it issued the next)
            // Loads under a condition do not do either: the compiler cannot tell that the conditional use has consumed the previous
            // turn's reply and waits before it reuses the registers.  So every turn loads all eight words, the finished end points
            // row 0 (one address for the whole wavefront).
            uint2 raw[8];
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                const P iF = (act >> (2 * c) & 1u) ? fRun[c] + 1 : (P)0, iL = (act >> (2 * c + 1) & 1u) ? lRun[c] + 1 : (P)0;
                raw[2 * c] = *reinterpret_cast<const uint2*>(t.rows + iF);
                raw[2 * c + 1] = *reinterpret_cast<const uint2*>(t.rows + iL);
            }
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                if (act >> (2 * c) & 1u) {
                    rows++;
                    const P nx = rowInT<P>(raw[2 * c]);
                    if (nx <= fOut[c]) fRun[c]++;
                    else act &= ~(1u << (2 * c));
                }
                if (act >> (2 * c + 1) & 1u) {
                    rows++;
                    const P nx = rowInT<P>(raw[2 * c + 1]);
                    if (nx <= lOut[c]) lRun[c]++;
                    else act &= ~(1u << (2 * c + 1));
                }
            }
        }
    }
    const P parentWidth = trivial.end - trivial.begin;
    // the toehold samples of the children that narrow the range (BMove::computeToehold / computeToeholdRev, bmove.cpp:222-266: the last
    // run of the range that holds the character): requested together, before any is used
    P smp[4];
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) { // (unconditional loads, see above: a child that needs no sample reads samplesFirst[0])
        const bool wanted = (found >> (c + 1) & 1u) && (need >> c & 1u) && lOut[c] + 1 - fOut[c] != parentWidth;
        const bool atEnd = !wanted || lSrc[c] == trivial.endRun;
        const uint64_t* sp = atEnd ? t.samplesFirst : t.samplesLast;
        smp[c] = (P)sp[wanted ? (atEnd ? trivial.endRun : lSrc[c]) : (P)0];
    }
    // MoveLFReprBP::getCumulativeCounts (moverepr.cpp:347-365): the '$' of the range, then the smaller characters
    P cum = (trivial.begin <= t.zeroCharPos && trivial.end > t.zeroCharPos) ? 1 : 0;
    uint32_t mask = 0;
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        MvPairT<P> one;
        MvPairT<P>& ch = PACK ? one : child[c];
        if (!(found >> (c + 1) & 1u)) { // addChar: setEmpty() (moverepr.cpp:313-316), SARangePair(range1, range1, 0, false, 0)
            ch.sa = {0, 0, 0, 0, false};
            ch.rev = ch.sa;
            ch.toehold = 0, ch.repEnd = false, ch.depth = 0;
            continue;
        }
        mask |= 1u << c;
        const P width = lOut[c] + 1 - fOut[c]; // = countChar(trivial, c + 1)
        if (need >> c & 1u) {
            const MvRangeT<P> range1 = {fOut[c], (P)(lOut[c] + 1), fRun[c], lRun[c], true};
            MvRangeT<P> second;
            const MvRangeT<P> noRange{0, 0, 0, 0, true};
            if (width == parentWidth) { // the other range and the toehold carry over
                second = selRange(mode == 2, noRange, other);
                ch.toehold = fw ? parent.toehold + (parent.repEnd ? 1 : 0) : parent.toehold - (parent.repEnd ? 0 : 1);
                ch.repEnd = parent.repEnd;
            } else {
                const MvRangeT<P> narrowed{(P)(other.begin + cum), (P)(other.begin + cum + width), other.beginRun, other.endRun, false};
                second = selRange(mode == 2, noRange, narrowed);
                ch.toehold = fw ? (P)(ix.n - 1) - (smp[c] - 1) : smp[c] - 1;
                ch.repEnd = fw;
            }
            ch.sa = selRange(fw, second, range1);
            ch.rev = selRange(fw, range1, second);
            ch.depth = parent.depth + 1;
            if (PACK) MvTraits::pack(ch, pk[c][0], pk[c][1], pk[c][2]);
        }
        cum += width;
    }
    return mask;
}

// The children of a parent IN SLOTS (round 4, the frontier kernel's instance on 32-bit positions).  moveChildrenCounted above keeps
// everything that is indexed by character in registers, unrolled over the four characters: a node has 1.1 children on average, so
// three quarters of what a wavefront issues there — LF images, widths, ranges, packing, and afterwards the matrix rows and the stores of
// the frontier kernel — is work on children that do not exist.  Here the scan's results live in LDS, [field][character][lane], written
// with the character a row holds as a run-time index, and the children are materialised ONE PER TURN in ascending character order into
// slot j = the lane's j-th existing child (static register indices; the loops over slots end as soon as no lane of the wavefront has a
// j-th child): `chars` holds the character of every slot (2 bits each), the return value is the number of children.
struct MvScanLds {
    uint32_t fOut[4][256], fRun[4][256], lOut[4][256], lRun[4][256], lSrc[4][256], smp[4][256];
};
__device__ __forceinline__ uint32_t moveChildrenSlots(const MoveDev& ix, const int mode, const MvPairT<uint32_t>& parent, uint32_t& rows, MvScanLds& L,
                                                      const uint32_t tid, uint4 (*pk)[3], uint32_t& chars) {
    typedef uint32_t P;
    const bool fw = mode == 0;
    MoveTable t;
    t.rows = fw ? ix.rev.rows : ix.fwd.rows;
    t.runs = fw ? ix.rev.runs : ix.fwd.runs;
    t.zeroCharPos = fw ? ix.rev.zeroCharPos : ix.fwd.zeroCharPos;
    t.samplesFirst = fw ? ix.rev.samplesFirst : ix.fwd.samplesFirst;
    t.samplesLast = fw ? ix.rev.samplesLast : ix.fwd.samplesLast;
    MvRangeT<P> trivial = selRange(fw, parent.rev, parent.sa);
    const MvRangeT<P> other = selRange(fw, parent.sa, parent.rev);
    if (!trivial.valid) { // (two binary searches between the enclosing run indices)
        P span = trivial.endRun - trivial.beginRun;
        while (span) {
            rows += 2;
            span >>= 1;
        }
        computeRunIndices(t, trivial);
    }
    uint32_t found = 0, seen = 0; // bits 0..4: character seen from the front / from the back
    uint32_t ffNeed = 0;          // end points (2 c: first, 2 c + 1: last) whose image left its target run
    {
        P runF = trivial.beginRun, posF = trivial.begin, runB = trivial.endRun, posB = trivial.end - 1;
        uint4 rowF = t.rows[runF], rowB = t.rows[runB];
        uint4 preF = t.rows[runF < trivial.endRun ? runF + 1 : runF], preB = t.rows[runB > trivial.beginRun ? runB - 1 : runB];
        rows += 2;
        bool fDone = false, bDone = false;
        while (true) {
            if (!fDone) {
                const uint32_t h = rowHead(rowF);
                if (!(found >> h & 1u)) {
                    found |= 1u << h;
                    if (h != 0u) {
                        const MoveRowT<P> r = unpackMoveRowT<P>(rowF);
                        const P off = posF - r.in;
                        const uint32_t gap = rowGap(rowF);
                        const bool ff = off >= gap; // (the row's gap field, move_dev.hpp: most images stay in the target run)
                        L.fOut[h - 1u][tid] = r.out + off;
                        L.fRun[h - 1u][tid] = r.outRun + ((ff && gap < MV_GAP_MAX) ? 1u : 0u);
                        ffNeed |= ff ? 1u << (2u * (h - 1u)) : 0u;
                    }
                }
                fDone = (found & 0x1Eu) == 0x1Eu || runF == trivial.endRun;
            }
            if (!bDone) {
                const uint32_t h = rowHead(rowB);
                if (!(seen >> h & 1u)) {
                    seen |= 1u << h;
                    if (h != 0u) {
                        const MoveRowT<P> r = unpackMoveRowT<P>(rowB);
                        const P off = posB - r.in;
                        const uint32_t gap = rowGap(rowB);
                        const bool ff = off >= gap;
                        L.lOut[h - 1u][tid] = r.out + off;
                        L.lRun[h - 1u][tid] = r.outRun + ((ff && gap < MV_GAP_MAX) ? 1u : 0u);
                        L.lSrc[h - 1u][tid] = runB;
                        ffNeed |= ff ? 2u << (2u * (h - 1u)) : 0u;
                    }
                }
                bDone = (seen & 0x1Eu) == 0x1Eu || runB == trivial.beginRun || (fDone && (seen & 0x1Eu) == (found & 0x1Eu));
            }
            if (fDone && bDone) break;
            const P posBNext = rowInT<P>(rowB) - 1;
            if (!fDone) {
                runF++;
                rowF = preF;
                rows++;
                posF = rowInT<P>(rowF);
            }
            if (!bDone) {
                posB = posBNext;
                runB--;
                rowB = preB;
                rows++;
            }
            preF = t.rows[runF < trivial.endRun ? runF + 1 : runF];
            preB = t.rows[runB > trivial.beginRun ? runB - 1 : runB];
        }
        found &= seen | 1u;
    }
    const uint32_t exist = (found >> 1) & 0xFu; // bit c: character c + 1 occurs in the range
    // fast-forwards (moverepr.cpp:283-293) of the end points whose image left its target run: all of them step together
    {
        uint32_t act = 0;
#pragma unroll
        for (uint32_t c = 0; c < 4; c++)
            if (exist >> c & 1u) act |= 3u << (2 * c);
        act &= ffNeed;
        while (act) {
            uint2 raw[8];
            P run[8];
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) { // (unconditional loads: a finished end point reads row 0, see moveChildrenCounted)
                run[2 * c] = (act >> (2 * c) & 1u) ? L.fRun[c][tid] : 0u;
                run[2 * c + 1] = (act >> (2 * c + 1) & 1u) ? L.lRun[c][tid] : 0u;
                raw[2 * c] = *reinterpret_cast<const uint2*>(t.rows + ((act >> (2 * c) & 1u) ? run[2 * c] + 1u : 0u));
                raw[2 * c + 1] = *reinterpret_cast<const uint2*>(t.rows + ((act >> (2 * c + 1) & 1u) ? run[2 * c + 1] + 1u : 0u));
            }
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                if (act >> (2 * c) & 1u) {
                    rows++;
                    if (rowInT<P>(raw[2 * c]) <= L.fOut[c][tid]) L.fRun[c][tid] = run[2 * c] + 1u;
                    else act &= ~(1u << (2 * c));
                }
                if (act >> (2 * c + 1) & 1u) {
                    rows++;
                    if (rowInT<P>(raw[2 * c + 1]) <= L.lOut[c][tid]) L.lRun[c][tid] = run[2 * c + 1] + 1u;
                    else act &= ~(1u << (2 * c + 1));
                }
            }
        }
    }
    const P parentWidth = trivial.end - trivial.begin;
    // the toehold samples of the children that narrow the range (bmove.cpp:222-266): requested together, before any is used
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) { // (unconditional loads: a child that needs no sample reads samplesFirst[0])
        const bool wanted = (exist >> c & 1u) && L.lOut[c][tid] + 1u - L.fOut[c][tid] != parentWidth;
        const P src = wanted ? L.lSrc[c][tid] : 0u;
        const bool atEnd = !wanted || src == trivial.endRun;
        const uint64_t* sp = atEnd ? t.samplesFirst : t.samplesLast;
        L.smp[c][tid] = (P)sp[wanted ? (atEnd ? trivial.endRun : src) : (P)0];
    }
    // MoveLFReprBP::getCumulativeCounts (moverepr.cpp:347-365): the '$' of the range, then the smaller characters
    P cum = (trivial.begin <= t.zeroCharPos && trivial.end > t.zeroCharPos) ? 1 : 0;
    uint32_t rem = exist, n = 0;
    chars = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        if (__ballot(rem != 0u) == 0ull) break; // (wave-uniform: no lane has a j-th child)
        if (rem) {
            const uint32_t c = (uint32_t)__ffs((int)rem) - 1u;
            rem &= rem - 1u;
            chars |= c << (2u * j);
            n++;
            const P fO = L.fOut[c][tid], lO = L.lOut[c][tid];
            const P width = lO + 1u - fO; // = countChar(trivial, c + 1)
            const MvRangeT<P> range1 = {fO, (P)(lO + 1u), L.fRun[c][tid], L.lRun[c][tid], true};
            const MvRangeT<P> noRange{0, 0, 0, 0, true};
            MvPairT<P> ch;
            MvRangeT<P> second;
            if (width == parentWidth) { // the other range and the toehold carry over
                second = selRange(mode == 2, noRange, other);
                ch.toehold = fw ? parent.toehold + (parent.repEnd ? 1 : 0) : parent.toehold - (parent.repEnd ? 0 : 1);
                ch.repEnd = parent.repEnd;
            } else {
                const MvRangeT<P> narrowed{(P)(other.begin + cum), (P)(other.begin + cum + width), other.beginRun, other.endRun, false};
                second = selRange(mode == 2, noRange, narrowed);
                const P sm = L.smp[c][tid];
                ch.toehold = fw ? (P)(ix.n - 1) - (sm - 1) : sm - 1;
                ch.repEnd = fw;
            }
            ch.sa = selRange(fw, second, range1);
            ch.rev = selRange(fw, range1, second);
            ch.depth = parent.depth + 1;
            MvTraits::pack(ch, pk[j][0], pk[j][1], pk[j][2]);
            cum += width;
        }
    }
    return n;
}

// IndexInterface::addChar (indexinterface.cpp:1034-1049) on this backend: code 1..4 extends `r` in `mode`, anything else
// (N) empties it without touching a counter.  Returns true if the range is still non-empty.
struct MvCounters {
    uint32_t nodes = 0, expansions = 0, rows = 0, started = 0;
};
__device__ __forceinline__ bool mvAddChar(const MoveDev& ix, int mode, uint32_t code, MvPair& r, MvCounters& c) {
    if (code >= 1 && code <= 4) {
        c.expansions++;
        // an empty range has no children (matchStringBidirectionally keeps calling addChar on the empty ranges of a k-mer
        // with an N or one that does not occur: the reference's walks then run on the range (0, 0) of run 0 and come back
        // empty as well, moverepr.cpp:309-327 — the extension is counted, nothing is walked here)
        if (MvTraits::empty(r)) {
            r = MvTraits::none();
            return false;
        }
        MvPair ch[4];
        const uint32_t mask = moveChildrenCounted(ix, mode, r, ch, c.rows, 1u << (code - 1));
        if (mask >> (code - 1) & 1u) {
            // (assigned under a condition per child, not selected with ?: — a select between objects keeps them in scratch memory)
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (code == j + 1) r = ch[j];
            c.nodes++;
            return true;
        }
    }
    r = MvTraits::none();
    return false;
}

__device__ inline MvPair mvCompleteRange(const MoveDev& ix) { // BMove::getCompleteRange (bmove.h:369-373)
    MvPair p;
    p.sa = {0, ix.n, 0, ix.fwd.runs - 1, true};
    p.rev = {0, ix.n, 0, ix.rev.runs - 1, true};
    p.toehold = ix.fwd.samplesLast[ix.fwd.runs - 1] - 1, p.repEnd = false, p.depth = 0;
    return p;
}

// ------------------------------------------------------------------ read preparation
// One thread per read: character codes of both strands (1..4 = ACGT, 5 = anything else: reads.h:43-58, nucleotide.h:250) and
// the eight match bit-strings of the read (dev_matrix.hpp: gString — read / reversed read x A, C, G, T; G zeroed beforehand).
__global__ void k_mvs_prep(const uint8_t* __restrict__ reads, const uint64_t* __restrict__ offs, uint32_t nReads, uint32_t maxLen, uint32_t gw,
                           uint8_t* __restrict__ seq, uint32_t* __restrict__ G) {
    // one thread per read and 32-character word (round 4; one thread per read with a read-modify-write of global memory per character
    // took 23 ms and 123 GB of traffic for the 250 MB of reads of BASELINE configs[4]): the thread owns word w of all eight bit-strings —
    // those of the read from the characters [32 w, 32 w + 32), those of the reversed read from the same positions counted from the end —
    // and the codes of both strands at those positions; words beyond the read are written as zeros.
    const uint64_t nW = (uint64_t)nReads * gw;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nW; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(t / gw), w = (uint32_t)(t % gw);
        const uint8_t* rd = reads + offs[r];
        const uint32_t len = min((uint32_t)(offs[r + 1] - offs[r]), maxLen);
        uint8_t* sF = seq + (size_t)(2 * r) * maxLen;
        uint8_t* sR = seq + (size_t)(2 * r + 1) * maxLen;
        uint32_t fw[4] = {0, 0, 0, 0}, rv[4] = {0, 0, 0, 0};
        for (uint32_t bb = 0; bb < 32; bb++) {
            const uint32_t i = 32u * w + bb;
            if (i >= len) break;
            const uint8_t a = rd[i] & 0xDF, a2 = rd[len - 1 - i] & 0xDF; // (the reversed read holds character len - 1 - i at position i)
            const uint32_t c = a == 'A' ? 1 : a == 'C' ? 2 : a == 'G' ? 3 : a == 'T' ? 4 : 5;
            const uint32_t c2 = a2 == 'A' ? 1 : a2 == 'C' ? 2 : a2 == 'G' ? 3 : a2 == 'T' ? 4 : 5;
            sF[i] = (uint8_t)c;
            sR[i] = (uint8_t)(c2 <= 4 ? 5 - c2 : 5);
#pragma unroll
            for (uint32_t k4 = 0; k4 < 4; k4++) {
                fw[k4] |= c == k4 + 1 ? 1u << bb : 0u;
                rv[k4] |= c2 == k4 + 1 ? 1u << bb : 0u;
            }
        }
        uint32_t* g = G + (size_t)r * 8 * gw;
#pragma unroll
        for (uint32_t k4 = 0; k4 < 4; k4++) {
            g[k4 * gw + w] = fw[k4];
            g[(4 + k4) * gw + w] = rv[k4];
        }
    }
}

// ------------------------------------------------------------------ prologue: partitioning + scheme selection
// SearchStrategy::partition (searchstrategy.cpp:141-419) and MultipleSchemes::createSearches (searchstrategy.h:2505-2537), one
// lane per read x strand.  Partition state in LDS ([field][part][lane]).
// (Register budget: the dynamic partitioning at 8 parts came out at 257 registers — one over the line for two wavefronts per SIMD — and ran one;
// every lane is a chain of dependent table rows, so wavefronts are what hides them: CMB_MVS_PARTS_WAVES per SIMD for the narrow tables.)
#ifndef CMB_MVS_PARTS_WAVES
#define CMB_MVS_PARTS_WAVES 2
#endif
template <int PARTITION, int MP = MAXP>
__global__ void __launch_bounds__(64, MP == MAXP ? CMB_MVS_PARTS_WAVES : 1)
k_mvs_parts(MvSearchIndex sx, const DevStrategyKT<MP>* __restrict__ stp, uint32_t nReads, uint32_t maxLen, const uint8_t* __restrict__ seqAll,
            const uint64_t* __restrict__ offs, PartOutT<MP>* __restrict__ partsOut, MoveRangeRec* __restrict__ exr, uint8_t* __restrict__ psel, Queues q) {
    __shared__ uint32_t pbe[MP][64];
    __shared__ unsigned long long wid[MP][64];
    // the exact-match range pair of every part while the read is partitioned: [part][plane][lane], packed as in the frontier's
    // records (3 x 16 bytes).  Dynamic partitioning extends a different part at almost every step; with the pairs in global memory
    // each step was a scattered 80-byte load and store around its row fetches (k_partition: 528 GB of traffic per 10^6-read step).
    extern __shared__ uint4 exLds[];
    const DevStrategyKT<MP>& st = *stp;
    const MoveDev& ix = sx.d;
    const uint32_t lane = threadIdx.x, total = 2 * nReads;
    MvCounters cnt;
    uint32_t flags = 0;
    for (uint32_t rs = blockIdx.x * 64u + lane; rs < total; rs += gridDim.x * 64u) {
        const uint32_t len = (uint32_t)(offs[(rs >> 1) + 1] - offs[rs >> 1]);
        const uint8_t* seq = seqAll + (size_t)rs * maxLen;
        const int P = st.numParts;
        psel[rs] = 0;
        if (P >= (int)len || P == 1 || len > (uint32_t)MAX_READ) { // naive fallback of the reference (searchstrategy.cpp:148-152)
            flags |= FLAG_UNSUPPORTED_READ;
            psel[rs] = 0x80u;
            continue;
        }
        auto PB = [&](int i) -> uint32_t { return pbe[i][lane] & 0xFFFFu; };
        auto PE = [&](int i) -> uint32_t { return pbe[i][lane] >> 16; };
        auto setPBE = [&](int i, uint32_t b, uint32_t e) { pbe[i][lane] = (b & 0xFFFFu) | (e << 16); };
        auto putEx = [&](int i, const MvPair& r) { MvTraits::pack(r, exLds[(i * 3 + 0) * 64 + lane], exLds[(i * 3 + 1) * 64 + lane], exLds[(i * 3 + 2) * 64 + lane]); };
        auto getEx = [&](int i) -> MvPair { return MvTraits::unpack(exLds[(i * 3 + 0) * 64 + lane], exLds[(i * 3 + 1) * 64 + lane], exLds[(i * 3 + 2) * 64 + lane]); };
        auto kmer = [&](uint32_t begin, uint32_t end) -> MvPair { // lookUpInKmerTable (indexinterface.h:590-594)
            uint32_t key = 0;
            bool valid = true;
            for (uint32_t i = begin; i < end; i++)
                if (seq[i] > 4) valid = false;
            if (!valid) return MvTraits::none();
            for (uint32_t i = 0; i < sx.kmerSize; i++) key = (key << 2) | (uint32_t)(seq[begin + i] - 1);
            return loadPair(sx.kmer[key]);
        };
        const uint32_t ws = sx.kmerSize;
        if (PARTITION != 2) {
            if (PARTITION == 0) { // partitionUniform (:194-209)
                for (int i = 0; i < P; i++) {
                    const uint32_t b = (uint32_t)((i * 1.0 / P) * len);
                    uint32_t e = (uint32_t)(((i + 1) * 1.0 / P) * len);
                    setPBE(i, b, e > len ? len : e);
                }
                setPBE(P - 1, PB(P - 1), len);
            } else { // setParts (:221-238)
                const int pSize = (int)len;
                const double* bg = st.begins;
                setPBE(0, 0, (uint32_t)(bg[0] * pSize) & 0xFFFFu);
                for (int i = 0; i < P - 2; i++) setPBE(i + 1, (uint32_t)(bg[i] * pSize), (uint32_t)(bg[i + 1] * pSize) & 0xFFFFu);
                setPBE(P - 1, (uint32_t)(bg[P - 2] * pSize), len);
                for (int i = 0; i < P; i++)
                    if (PE(i) > len) setPBE(i, PB(i), len);
            }
            // calculateExactMatchRanges (:158-190): every part forwards, then the last part again backwards, uni-directionally
            for (int stage = 0; stage <= P; stage++) {
                const int i = stage < P ? stage : P - 1;
                const uint32_t b = PB(i), e = PE(i), size = e > b ? e - b : 0;
                MvPair r;
                if (stage < P) {
                    const uint32_t start = b + (size >= ws ? ws : 0);
                    r = size >= ws ? kmer(b, start) : mvCompleteRange(ix);
                    for (uint32_t j = start; j < e; j++)
                        if (!mvAddChar(ix, 0, seq[j], r, cnt)) break;
                } else {
                    const uint32_t end = size >= ws ? e - ws : e;
                    r = size >= ws ? kmer(end, e) : mvCompleteRange(ix);
                    for (uint32_t j = end; j-- > b;)
                        if (!mvAddChar(ix, 2, seq[j], r, cnt)) break;
                }
                putEx(i, r);
                wid[i][lane] = r.sa.end > r.sa.begin ? r.sa.end - r.sa.begin : 0;
            }
        } else { // seed (:381-419) + partitionDynamic (:299-379)
            const bool useKmer = ((uint32_t)P * ws < (len * 2) / 3) && (len >= st.kmerCutOff);
            const int wSize = useKmer ? (int)ws : 1;
            setPBE(0, 0, (uint32_t)wSize);
            for (int i = 1; i < P - 1; i++) {
                const uint32_t b = (uint32_t)(uint16_t)(int)((st.seeding[i - 1] * len) - (wSize / 2));
                setPBE(i, b, (b + wSize) & 0xFFFFu);
            }
            setPBE(P - 1, len - wSize, len);
            bool overlap = false;
            for (int i = 0; i + 1 < P; i++)
                if (PE(i) > PB(i + 1)) overlap = true; // `assert(parts[i].end() <= parts[i + 1].begin())` (:404-407)
            if (overlap) {
                flags |= FLAG_SEED_OVERLAP;
                psel[rs] = 0x80u;
                continue;
            }
            for (int i = 0; i < P; i++) {
                MvPair r;
                if (useKmer) r = kmer(PB(i), PE(i));
                else { // BMove::getRangeOfSingleChar (bmove.cpp:484-497): the complete range extended backward, no counters
                    const uint32_t code = seq[PB(i)];
                    r = MvTraits::none();
                    if (code >= 1 && code <= 4) {
                        MvPair ch[4];
                        uint32_t rows = 0;
                        const uint32_t mask = moveChildrenCounted(ix, 1, mvCompleteRange(ix), ch, rows, 1u << (code - 1));
                        if (mask >> (code - 1) & 1u) {
#pragma unroll
                            for (uint32_t j = 0; j < 4; j++)
                                if (code == j + 1) r = ch[j];
                        }
                    }
                }
                putEx(i, r);
                wid[i][lane] = r.sa.end > r.sa.begin ? r.sa.end - r.sa.begin : 0;
            }
            int partToExtend = 0, dynDir = 0;
            for (uint32_t j = (uint32_t)(P * wSize); j < len; j++) {
                unsigned long long maxRangeWeighted = 0;
                for (int i = 0; i < P; i++) {
                    const bool noLeft = (i == 0) || PB(i) == PE(i - 1);
                    const bool noRight = (i == P - 1) || PE(i) == PB(i + 1);
                    if (noLeft && noRight) continue;
                    const unsigned long long wv = wid[i][lane] * st.weights[i];
                    if (wv > maxRangeWeighted) {
                        maxRangeWeighted = wv;
                        partToExtend = i;
                        if (noLeft) dynDir = 0;
                        else if (noRight) dynDir = 1;
                        else dynDir = (wid[i - 1][lane] < wid[i + 1][lane]) ? 1 : 0;
                    }
                }
                if (maxRangeWeighted == 0) { // extendParts (:283-297)
                    for (int i = 0; i < P; i++) {
                        if (i != P - 1 && PE(i) != PB(i + 1)) setPBE(i, PB(i), PB(i + 1));
                        if (i != 0 && PB(i) != PE(i - 1)) setPBE(i, PE(i - 1), PE(i));
                    }
                    break;
                }
                uint32_t code;
                if (dynDir == 0) {
                    setPBE(partToExtend, PB(partToExtend), PE(partToExtend) + 1);
                    code = seq[PE(partToExtend) - 1];
                } else {
                    setPBE(partToExtend, PB(partToExtend) - 1, PE(partToExtend));
                    code = seq[PB(partToExtend)];
                }
                MvPair r = getEx(partToExtend);
                (void)mvAddChar(ix, partToExtend == P - 1 ? 2 : dynDir, code, r, cnt);
                putEx(partToExtend, r);
                wid[partToExtend][lane] = r.sa.end > r.sa.begin ? r.sa.end - r.sa.begin : 0;
            }
        }
        // createSearches (searchstrategy.h:2505-2537): sums and comparisons in the reference's 32-bit `unsigned int`
        int sel = 0;
        if (st.nSchemes > 1) {
            uint32_t tot = 0;
            for (int i = 0; i < P; i++) tot += (uint32_t)wid[i][lane];
            if (tot > (uint32_t)P) {
                uint32_t minValue = (uint32_t)wid[st.sch[0].critical][lane];
                for (int i = 1; i < st.nSchemes; i++) {
                    const uint32_t w = (uint32_t)wid[st.sch[i].critical][lane];
                    if (w < minValue) {
                        minValue = w;
                        sel = i;
                    }
                }
            }
        }
        PartOutT<MP> po;
#pragma unroll
        for (int i = 0; i < MP; i++) {
            po.pb[i] = i < P ? (uint16_t)PB(i) : (uint16_t)0;
            po.pe[i] = i < P ? (uint16_t)PE(i) : (uint16_t)0;
        }
        partsOut[rs] = po;
        psel[rs] = (uint8_t)sel;
        for (int i = 0; i < P; i++) exr[(size_t)i * total + rs] = storePair(getEx(i)); // (k_mvs_exact starts from these)
    }
    const uint32_t local[3] = {cnt.nodes, cnt.expansions, cnt.rows};
    const int which[3] = {0, 7, 13};
    flushCounters(q, local, which, 3);
    if (flags) atomicOr(&q.cnt[3], flags);
}

// ------------------------------------------------------------------ prologue: exact phases of every search
// SearchStrategy::doRecSearch (searchstrategy.cpp:1181-1254) up to the first approximate phase, one lane per (read x strand,
// search of the selected scheme).  Emits one MvTask per search that starts (recApproxMatchEditEntry of this flavour counts
// every one of them as SEARCH_STARTED, indexinterface.cpp:1321-1323).
template <int MP = MAXP>
__global__ void __launch_bounds__(64)
k_mvs_exact(MvSearchIndex sx, const DevStrategyKT<MP>* __restrict__ stp, uint32_t nReads, uint32_t maxLen, uint32_t nSlots,
            const uint8_t* __restrict__ seqAll, const PartOutT<MP>* __restrict__ parts, const MoveRangeRec* __restrict__ exr,
            const uint8_t* __restrict__ psel, MvTask* __restrict__ tasks, uint32_t taskCap, Queues q) {
    const DevStrategyKT<MP>& st = *stp;
    const MoveDev& ix = sx.d;
    const uint32_t total = 2 * nReads;
    const uint64_t nWork = (uint64_t)total * nSlots;
    MvCounters cnt;
    uint32_t flags = 0;
    for (uint64_t w = blockIdx.x * 64ull + threadIdx.x; w < nWork; w += (uint64_t)gridDim.x * 64ull) {
        const uint32_t rs = (uint32_t)(w / nSlots), slot = (uint32_t)(w % nSlots);
        const uint8_t ps = psel[rs];
        if (ps & 0x80u) continue;
        const DevSchemeT<MP>& sch = st.sch[ps];
        if (slot >= sch.nSearches) continue;
        const DevSearchT<MP>& s = sch.s[slot];
        const uint8_t* seq = seqAll + (size_t)rs * maxLen;
        const PartOutT<MP> po = parts[rs];
        MvPair cur;
        uint32_t idx = 0, depth = 0;
        if (s.U[0] > 0) { // the first part already allows errors: start from the empty match
            cur = mvCompleteRange(ix);
        } else {
            const int first = s.order[0];
            cur = loadPair(exr[(size_t)first * total + rs]);
            if (MvTraits::empty(cur)) continue; // `startRange.width() > index.getSwitchPoint()` with switch point 0
            uint32_t p = 1;
            depth = (uint32_t)po.pe[first] - po.pb[first];
            bool alive = true;
            while (s.U[p] == 0) {
                const int part = s.order[p];
                const uint32_t b = po.pb[part], e = po.pe[part], n = e > b ? e - b : 0;
                const bool uni = s.uniAll || p >= (uint32_t)s.uniIdx;
                const int mode = uni ? 2 : (s.dir[p] == 0 ? 0 : 1);
                for (uint32_t ci = 0; ci < n; ci++) {
                    const uint32_t code = seq[s.dir[p] == 0 ? b + ci : e - ci - 1];
                    if (!mvAddChar(ix, mode, code, cur, cnt)) break;
                }
                if (MvTraits::empty(cur)) {
                    alive = false;
                    break;
                }
                depth += n;
                p++;
            }
            if (!alive) continue;
            idx = p;
        }
        if (st.metric == 1) cnt.started++;
        const uint32_t o = atomicAdd(&q.cnt[5], 1u);
        if (o >= taskCap) {
            flags |= FLAG_DFS_OVERFLOW;
            continue;
        }
        MvTask t;
        t.rsId = rs, t.scheme = ps, t.search = (uint8_t)slot, t.idx = (uint8_t)idx, t.pad = 0, t.depth = depth, t.pad2 = 0;
        t.r = storePair(cur);
        tasks[o] = t;
    }
    const uint32_t local[4] = {cnt.nodes, cnt.expansions, cnt.rows, cnt.started};
    const int which[4] = {0, 7, 13, 6};
    flushCounters(q, local, which, 4);
    if (flags) atomicOr(&q.cnt[3], flags);
}

// ------------------------------------------------------------------ the frontier: one level per launch
// Expansion of a frontier node (extendFMPos + branchAndBound + the stack loop of recApproxMatchEdit, indexinterface.cpp:506-561,
// :675-697, without the in-text switch this flavour does not have).  Node = range pair (3 planes) + the three planes of
// dev_bfs_edit.hpp: {row | score << 16, ctx, fc, RAC bit | mode << 8} {HP, HN} {final-column distances}.
// Geo: the record geometry of dev_bfs_edit.hpp — GeoN32 (up to 7 errors, the instance of BASELINE configs[4]: the in-index matrix on
// 32-bit words since round 4 up to 6 errors; GeoN, the reference's 64-bit words, at 7 errors and for a batch one of whose phases does not fit the small matrix), GeoW (8 ... 10),
// GeoX (11 ... 13: the in-index matrix with 16-row blocks).
template <class Geo = GeoN, typename P = uint64_t>
__device__ __forceinline__ void mvExpand(const MoveDev& ix, const MvBufs& B, uint32_t pass, const Queues& q, uint32_t bid, uint32_t nBlocks) {
    __shared__ uint32_t sh[4][5];
    // per-lane state that is touched once per expansion lives in LDS, [field][lane], not in registers (the kernel waits for dependent row
    // fetches: what it can keep in flight is set by its registers): the four match words of the row block
    __shared__ uint64_t ldsM[4][256];
    const uint32_t tid = threadIdx.x;
    typedef typename Geo::W W; // the word of a matrix row (dev_bfs_edit.hpp: 64 bits, or 32 for GeoN32)
    typedef typename Geo::Pack EdPack;
    constexpr uint32_t ED_CELLS = Geo::CELLS, ED_MAX = Geo::ED_MAX, EV_U4 = 1u + Geo::PK_U4;
    constexpr uint32_t PU = MvTraits::PAIR_U4, FU = PU + 1;
    const uint32_t nIn = min(B.nq[pass], B.qCap);
    const uint4* __restrict__ Qi = B.Q[pass & 1u];
    uint4* __restrict__ Qo = B.Q[(pass + 1u) & 1u];
    uint4* __restrict__ Eo = B.Ev[(pass + 1u) & 1u];
    const uint32_t qCap = B.qCap;
    uint32_t flags = 0;
    uint32_t cChildren = 0, cExp = 0, cRows = 0; // (per lane and launch: far below 2^32)
    for (uint32_t base = bid * 256u; base < nIn; base += nBlocks * 256u) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        const bool act = i < nIn;
        uint32_t kinds = 0; // 4 bits per child: kind | needF << 2
        uint32_t row1 = 0, ctx = 0, fcP = BFS_NONE;
        uint4 pk[4][3]; // the children's range pairs, packed as they are stored (twelve registers each instead of twenty-one)
        W cHP[4], cHN[4];
        uint32_t cMeta[4]; // score << 16 | RAC bit << 8 | final-column distance
        uint32_t hotY = 0, clSize = 0; // (the band geometry stays packed as the context's hot word holds it)
        int md = 0;
        MvPairT<P> parent{};
        uint32_t row = 0, score = 0, pRac = 0, blk = 0;
        W pHP = 0, pHN = 0;
        const uint4* Cx = B.C;
        if (act) {
            const uint4 n1 = qLoad(Qi + (size_t)PU * qCap + i), n2 = qLoad(Qi + (size_t)(PU + 1) * qCap + i);
            parent = MvTraits::unpackT<P>(qLoad(Qi + i), qLoad(Qi + (size_t)qCap + i), qLoad(Qi + (size_t)2 * qCap + i));
            ctx = n1.y;
            fcP = n1.z;
            row = n1.x & 0xFFFFu;
            score = n1.x >> 16;
            md = (int)((n1.w >> 8) & 3u);
            Cx = B.C + (size_t)CMB_IDX(ctx, B.cCap, 1) * B.ctxU4;
            blk = (row + 1) / Geo::CTX_BLOCK;
            const uint4 hot = Cx[CTX_HOT];
            const uint4 mA = Cx[CTX_M + 2 * blk], mB = Cx[CTX_M + 1 + 2 * blk];
            ldsM[0][tid] = u64of(mA.x, mA.y), ldsM[1][tid] = u64of(mA.z, mA.w), ldsM[2][tid] = u64of(mB.x, mB.y), ldsM[3][tid] = u64of(mB.z, mB.w);
            hotY = hot.y;
            clSize = hot.w >> 23;
            Geo::unpackRow(n2, pHP, pHN);
            pRac = n1.w & 63u;
        }
        // ---- walk: an expansion that yields exactly one plain node (outside the final column) is followed at once by the
        // expansion of that child, by the same lane — no node record written and read back, no queue slot — for up to B.chain
        // rows; on this index most of a search is such a path (no in-text switch ends it: ranges stay narrow down to the last
        // row).  The lane stops at the first expansion that produces anything else; its children are what is appended below.
        bool walking = act;
        for (uint32_t step = 0; step < B.chain; step++) { // (wave-uniform exit below)
            if (walking) {
                row1 = row + 1;
                if (row1 / Geo::CTX_BLOCK != blk) { // the walk crossed into the next block of match words
                    blk = row1 / Geo::CTX_BLOCK;
                    const uint4 mA = Cx[CTX_M + 2 * blk], mB = Cx[CTX_M + 1 + 2 * blk];
                    ldsM[0][tid] = u64of(mA.x, mA.y), ldsM[1][tid] = u64of(mA.z, mA.w), ldsM[2][tid] = u64of(mB.x, mB.y), ldsM[3][tid] = u64of(mB.z, mB.w);
                }
                uint32_t rows = 0, mask;
#pragma unroll
                for (uint32_t c = 0; c < 4; c++) { // (nothing of the previous expansion stays alive across the scan)
                    pk[c][0] = pk[c][1] = pk[c][2] = make_uint4(0, 0, 0, 0);
                    cHP[c] = cHN[c] = 0;
                    cMeta[c] = 0;
                }
                mask = moveChildrenCounted<true, P>(ix, md, parent, nullptr, rows, 0xFu, pk);
                cRows += rows;
                cExp++;
                MatGeom g;
                g.n = hotY & 0x1FFu;
                g.m = (hotY >> 9) & 0x1FFu;
                g.Wv = (hotY >> 18) & 31u;
                g.Wh = (hotY >> 23) & 15u;
                g.maxED = (hotY >> 27) & 15u;
                const bool inFC = g.inFinalColumn(row1);
                if (inFC && clSize + row1 - g.m >= ED_CELLS) flags |= FLAG_CAPACITY;
                kinds = 0;
#pragma unroll
                for (uint32_t c = 0; c < 4; c++) {
                    if (!(mask >> c & 1u)) continue;
                    cChildren++;
                    const W M = Geo::mword(ldsM[c][tid], row1);
                    W HP = pHP, HN = pHN, RAC = Geo::racBit(pRac), D0;
                    uint32_t sc = score;
                    const bool valid = Geo::row(g, row1, M, HP, HN, D0, RAC, sc);
                    if (!valid && !inFC) continue; // pruned when popped (branchAndBound returns true, :560)
                    uint32_t res = KIND_NODE, aux = 0;
                    if (inFC) {
                        const uint32_t ed = Geo::cell(row1, g.n - 1, HP, HN, sc);
                        aux = min(ed, ED_MAX);
                        res |= 4u;
                        if (ed > ED_MAX) flags |= FLAG_CAPACITY;
                        if (!valid || Geo::ovgl(g, row1, HN)) res = (res & ~3u) | KIND_EVENT;
                    }
                    kinds |= res << (4 * c);
                    cHP[c] = HP, cHN[c] = HN;
                    if (sc > 0xFFFFu) flags |= FLAG_CAPACITY;
                    cMeta[c] = (sc << 16) | (Geo::racIdx(RAC) << 8) | aux;
                }
                const bool single = kinds == (uint32_t)KIND_NODE || kinds == ((uint32_t)KIND_NODE << 4) || kinds == ((uint32_t)KIND_NODE << 8) ||
                                    kinds == ((uint32_t)KIND_NODE << 12);
                if (single && step + 1u < B.chain && row1 + 1u < B.ctxMblk * Geo::CTX_BLOCK) {
#pragma unroll
                    for (uint32_t c = 0; c < 4; c++)
                        if (kinds == ((uint32_t)KIND_NODE << (4 * c))) {
                            parent = MvTraits::unpackT<P>(pk[c][0], pk[c][1], pk[c][2]);
                            pHP = cHP[c];
                            pHN = cHN[c];
                            score = cMeta[c] >> 16;
                            pRac = (cMeta[c] >> 8) & 63u;
                        }
                    row = row1;
                    kinds = 0; // (nothing of this expansion is left to append)
                } else {
                    walking = false;
                }
            }
            if (__ballot(walking) == 0ull) break;
        }
        uint32_t nNode = 0, nEv = 0, nF = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t kd = (kinds >> (4 * c)) & 3u;
            nNode += kd == KIND_NODE;
            nEv += kd == KIND_EVENT;
            nF += (kinds >> (4 * c + 2)) & 1u;
        }
        const uint32_t want[4] = {nNode, nEv, 0u, nF};
        uint32_t got[4];
        blockAppend4(&B.nq[pass + 1], &B.ne[pass + 1], &q.cnt[0], &B.pool[0], want, sh, got);
        uint32_t oNode = got[0], oEv = got[1], oF = got[3];
        bool ok = true;
        if (oNode + nNode > qCap) { ok = false; flags |= FLAG_BFS_Q; }
        if (oEv + nEv > B.evCap) { ok = false; flags |= FLAG_BFS_EV; }
        if (oF + nF > B.fCap) { ok = false; flags |= FLAG_BFS_F; }
        if (kinds != 0u && ok) {
            const uint32_t cell = min(clSize + row1 - ((hotY >> 9) & 0x1FFu), ED_CELLS - 1u);
            EdPack pack{};
            if (fcP != BFS_NONE && (kinds & 0x4444u)) packLoad(Qi + (size_t)(PU + 2) * qCap + i, qCap, pack);
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                const uint32_t kd = (kinds >> (4 * c)) & 3u;
                if (kd == KIND_NONE) continue;
                const bool wantF = (kinds >> (4 * c + 2)) & 1u;
                uint32_t fc = BFS_NONE;
                if (wantF) {
                    fc = oF++;
                    uint4* Fr = B.F + (size_t)CMB_IDX(fc, B.fCap, 9) * FU;
                    Fr[0] = pk[c][0], Fr[1] = pk[c][1], Fr[2] = pk[c][2];
                    Fr[PU] = make_uint4(row1 | ((c + 1) << 16), fcP, 0u, 0u);
                }
                if (kd == KIND_NODE) {
                    const uint32_t o = oNode++;
                    qStore(Qo + o, pk[c][0]), qStore(Qo + (size_t)qCap + o, pk[c][1]), qStore(Qo + (size_t)2 * qCap + o, pk[c][2]);
                    qStore(Qo + (size_t)PU * qCap + o, make_uint4(row1 | (cMeta[c] & 0xFFFF0000u), ctx, fc, ((cMeta[c] >> 8) & 63u) | ((uint32_t)md << 8)));
                    qStore(Qo + (size_t)(PU + 1) * qCap + o, Geo::packRow(cHP[c], cHN[c]));
                    if (wantF) {
                        EdPack p2 = pack;
                        edPut(p2, cell, cMeta[c] & 0xFFu);
                        packStore(Qo + (size_t)(PU + 2) * qCap + o, qCap, p2);
                    }
                } else { // KIND_EVENT
                    EdPack p2 = pack;
                    edPut(p2, cell, cMeta[c] & 0xFFu);
                    Eo[(size_t)EV_U4 * oEv] = make_uint4(ctx, fc, 0xFFFFFFFFu, cell);
                    packStore(Eo + (size_t)EV_U4 * oEv + 1, 1, p2);
                    oEv++;
                }
            }
        }
    }
    // per-block counters (summed by k_mvs_finish): one writer per slot and launch, launches are ordered
    unsigned long long v[4] = {cChildren, cExp, cChildren, cRows}; // (every child gets its matrix row)
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[j] += __shfl_xor(v[j], d);
    }
    __shared__ unsigned long long shc[4][4];
    if ((threadIdx.x & 63u) == 0)
        for (int j = 0; j < 4; j++) shc[threadIdx.x >> 6][j] = v[j];
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long t = shc[0][threadIdx.x] + shc[1][threadIdx.x] + shc[2][threadIdx.x] + shc[3][threadIdx.x];
        if (t) B.blockCnt[(size_t)bid * 4 + threadIdx.x] += t;
    }
    if (flags) atomicOr(&q.cnt[3], flags);
}

// The same on 32-bit positions with the children IN SLOTS (moveChildrenSlots): slot j = the lane's j-th existing child; the loops over
// the children — matrix rows, the choice of the child a lane walks on with, the records — run over slots and end as soon as no lane of
// the wavefront has a j-th child (1.1 children per node: most steps take one or two turns where the loops over characters took four).
template <class Geo = GeoN>
__device__ __forceinline__ void mvExpandSlots(const MoveDev& ix, const MvBufs& B, uint32_t pass, const Queues& q, uint32_t bid, uint32_t nBlocks) {
    __shared__ uint32_t sh[4][5];
    // per-lane state that is touched once per expansion lives in LDS, [field][lane], not in registers (the kernel waits for dependent row
    // fetches: what it can keep in flight is set by its registers): the four match words of the row block
    __shared__ uint64_t ldsM[4][256];
    __shared__ MvScanLds scanLds; // the scan's results per character (moveChildrenSlots)
    typedef uint32_t P;
    const uint32_t tid = threadIdx.x;
    typedef typename Geo::W W; // the word of a matrix row (dev_bfs_edit.hpp: 64 bits, or 32 for GeoN32)
    typedef typename Geo::Pack EdPack;
    constexpr uint32_t ED_CELLS = Geo::CELLS, ED_MAX = Geo::ED_MAX, EV_U4 = 1u + Geo::PK_U4;
    constexpr uint32_t PU = MvTraits::PAIR_U4, FU = PU + 1;
    const uint32_t nIn = min(B.nq[pass], B.qCap);
    const uint4* __restrict__ Qi = B.Q[pass & 1u];
    uint4* __restrict__ Qo = B.Q[(pass + 1u) & 1u];
    uint4* __restrict__ Eo = B.Ev[(pass + 1u) & 1u];
    const uint32_t qCap = B.qCap;
    uint32_t flags = 0;
    uint32_t cChildren = 0, cExp = 0, cRows = 0; // (per lane and launch: far below 2^32)
    for (uint32_t base = bid * 256u; base < nIn; base += nBlocks * 256u) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        const bool act = i < nIn;
        uint32_t kinds = 0; // 4 bits per child: kind | needF << 2
        uint32_t row1 = 0, ctx = 0, fcP = BFS_NONE;
        uint4 pk[4][3]; // the children's range pairs BY SLOT, packed as they are stored
        uint32_t chars = 0, nCh = 0; // character of every slot (2 bits each), number of slots in use
        W cHP[4], cHN[4];
        uint32_t cMeta[4]; // score << 16 | RAC bit << 8 | final-column distance
        uint32_t hotY = 0, clSize = 0; // (the band geometry stays packed as the context's hot word holds it)
        int md = 0;
        MvPairT<P> parent{};
        uint32_t row = 0, score = 0, pRac = 0, blk = 0;
        W pHP = 0, pHN = 0;
        const uint4* Cx = B.C;
        if (act) {
            const uint4 n1 = qLoad(Qi + (size_t)PU * qCap + i), n2 = qLoad(Qi + (size_t)(PU + 1) * qCap + i);
            parent = MvTraits::unpackT<P>(qLoad(Qi + i), qLoad(Qi + (size_t)qCap + i), qLoad(Qi + (size_t)2 * qCap + i));
            ctx = n1.y;
            fcP = n1.z;
            row = n1.x & 0xFFFFu;
            score = n1.x >> 16;
            md = (int)((n1.w >> 8) & 3u);
            Cx = B.C + (size_t)CMB_IDX(ctx, B.cCap, 1) * B.ctxU4;
            blk = (row + 1) / Geo::CTX_BLOCK;
            const uint4 hot = Cx[CTX_HOT];
            const uint4 mA = Cx[CTX_M + 2 * blk], mB = Cx[CTX_M + 1 + 2 * blk];
            ldsM[0][tid] = u64of(mA.x, mA.y), ldsM[1][tid] = u64of(mA.z, mA.w), ldsM[2][tid] = u64of(mB.x, mB.y), ldsM[3][tid] = u64of(mB.z, mB.w);
            hotY = hot.y;
            clSize = hot.w >> 23;
            Geo::unpackRow(n2, pHP, pHN);
            pRac = n1.w & 63u;
        }
        // ---- walk: an expansion that yields exactly one plain node (outside the final column) is followed at once by the
        // expansion of that child, by the same lane — no node record written and read back, no queue slot — for up to B.chain
        // rows; on this index most of a search is such a path (no in-text switch ends it: ranges stay narrow down to the last
        // row).  The lane stops at the first expansion that produces anything else; its children are what is appended below.
        bool walking = act;
        for (uint32_t step = 0; step < B.chain; step++) { // (wave-uniform exit below)
            if (walking) {
                row1 = row + 1;
                if (row1 / Geo::CTX_BLOCK != blk) { // the walk crossed into the next block of match words
                    blk = row1 / Geo::CTX_BLOCK;
                    const uint4 mA = Cx[CTX_M + 2 * blk], mB = Cx[CTX_M + 1 + 2 * blk];
                    ldsM[0][tid] = u64of(mA.x, mA.y), ldsM[1][tid] = u64of(mA.z, mA.w), ldsM[2][tid] = u64of(mB.x, mB.y), ldsM[3][tid] = u64of(mB.z, mB.w);
                }
                uint32_t rows = 0;
#pragma unroll
                for (uint32_t c = 0; c < 4; c++) { // (nothing of the previous expansion stays alive across the scan)
                    pk[c][0] = pk[c][1] = pk[c][2] = make_uint4(0, 0, 0, 0);
                    cHP[c] = cHN[c] = 0;
                    cMeta[c] = 0;
                }
                nCh = moveChildrenSlots(ix, md, parent, rows, scanLds, tid, pk, chars);
                cRows += rows;
                cExp++;
                MatGeom g;
                g.n = hotY & 0x1FFu;
                g.m = (hotY >> 9) & 0x1FFu;
                g.Wv = (hotY >> 18) & 31u;
                g.Wh = (hotY >> 23) & 15u;
                g.maxED = (hotY >> 27) & 15u;
                const bool inFC = g.inFinalColumn(row1);
                if (inFC && clSize + row1 - g.m >= ED_CELLS) flags |= FLAG_CAPACITY;
                kinds = 0;
#pragma unroll
                for (uint32_t c = 0; c < 4; c++) { // (c: the slot)
                    if (__ballot(c < nCh) == 0ull) break; // (wave-uniform)
                    if (c >= nCh) continue;
                    cChildren++;
                    const W M = Geo::mword(ldsM[(chars >> (2u * c)) & 3u][tid], row1);
                    W HP = pHP, HN = pHN, RAC = Geo::racBit(pRac), D0;
                    uint32_t sc = score;
                    const bool valid = Geo::row(g, row1, M, HP, HN, D0, RAC, sc);
                    if (!valid && !inFC) continue; // pruned when popped (branchAndBound returns true, :560)
                    uint32_t res = KIND_NODE, aux = 0;
                    if (inFC) {
                        const uint32_t ed = Geo::cell(row1, g.n - 1, HP, HN, sc);
                        aux = min(ed, ED_MAX);
                        res |= 4u;
                        if (ed > ED_MAX) flags |= FLAG_CAPACITY;
                        if (!valid || Geo::ovgl(g, row1, HN)) res = (res & ~3u) | KIND_EVENT;
                    }
                    kinds |= res << (4 * c);
                    cHP[c] = HP, cHN[c] = HN;
                    if (sc > 0xFFFFu) flags |= FLAG_CAPACITY;
                    cMeta[c] = (sc << 16) | (Geo::racIdx(RAC) << 8) | aux;
                }
                const bool single = kinds == (uint32_t)KIND_NODE || kinds == ((uint32_t)KIND_NODE << 4) || kinds == ((uint32_t)KIND_NODE << 8) ||
                                    kinds == ((uint32_t)KIND_NODE << 12);
                if (single && step + 1u < B.chain && row1 + 1u < B.ctxMblk * Geo::CTX_BLOCK) {
#pragma unroll
                    for (uint32_t c = 0; c < 4; c++)
                        if (kinds == ((uint32_t)KIND_NODE << (4 * c))) {
                            parent = MvTraits::unpackT<P>(pk[c][0], pk[c][1], pk[c][2]);
                            pHP = cHP[c];
                            pHN = cHN[c];
                            score = cMeta[c] >> 16;
                            pRac = (cMeta[c] >> 8) & 63u;
                        }
                    row = row1;
                    kinds = 0; // (nothing of this expansion is left to append)
                } else {
                    walking = false;
                }
            }
            if (__ballot(walking) == 0ull) break;
        }
        uint32_t nNode = 0, nEv = 0, nF = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t kd = (kinds >> (4 * c)) & 3u;
            nNode += kd == KIND_NODE;
            nEv += kd == KIND_EVENT;
            nF += (kinds >> (4 * c + 2)) & 1u;
        }
        const uint32_t want[4] = {nNode, nEv, 0u, nF};
        uint32_t got[4];
        blockAppend4(&B.nq[pass + 1], &B.ne[pass + 1], &q.cnt[0], &B.pool[0], want, sh, got);
        uint32_t oNode = got[0], oEv = got[1], oF = got[3];
        bool ok = true;
        if (oNode + nNode > qCap) { ok = false; flags |= FLAG_BFS_Q; }
        if (oEv + nEv > B.evCap) { ok = false; flags |= FLAG_BFS_EV; }
        if (oF + nF > B.fCap) { ok = false; flags |= FLAG_BFS_F; }
        if (kinds != 0u && ok) {
            const uint32_t cell = min(clSize + row1 - ((hotY >> 9) & 0x1FFu), ED_CELLS - 1u);
            EdPack pack{};
            if (fcP != BFS_NONE && (kinds & 0x4444u)) packLoad(Qi + (size_t)(PU + 2) * qCap + i, qCap, pack);
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) { // (c: the slot)
                if (__ballot((kinds >> (4 * c)) != 0u) == 0ull) break; // (wave-uniform: no lane has a record in this slot or a later one)
                const uint32_t kd = (kinds >> (4 * c)) & 3u;
                if (kd == KIND_NONE) continue;
                const bool wantF = (kinds >> (4 * c + 2)) & 1u;
                uint32_t fc = BFS_NONE;
                if (wantF) {
                    fc = oF++;
                    uint4* Fr = B.F + (size_t)CMB_IDX(fc, B.fCap, 9) * FU;
                    Fr[0] = pk[c][0], Fr[1] = pk[c][1], Fr[2] = pk[c][2];
                    Fr[PU] = make_uint4(row1 | ((((chars >> (2u * c)) & 3u) + 1u) << 16), fcP, 0u, 0u);
                }
                if (kd == KIND_NODE) {
                    const uint32_t o = oNode++;
                    qStore(Qo + o, pk[c][0]), qStore(Qo + (size_t)qCap + o, pk[c][1]), qStore(Qo + (size_t)2 * qCap + o, pk[c][2]);
                    qStore(Qo + (size_t)PU * qCap + o, make_uint4(row1 | (cMeta[c] & 0xFFFF0000u), ctx, fc, ((cMeta[c] >> 8) & 63u) | ((uint32_t)md << 8)));
                    qStore(Qo + (size_t)(PU + 1) * qCap + o, Geo::packRow(cHP[c], cHN[c]));
                    if (wantF) {
                        EdPack p2 = pack;
                        edPut(p2, cell, cMeta[c] & 0xFFu);
                        packStore(Qo + (size_t)(PU + 2) * qCap + o, qCap, p2);
                    }
                } else { // KIND_EVENT
                    EdPack p2 = pack;
                    edPut(p2, cell, cMeta[c] & 0xFFu);
                    Eo[(size_t)EV_U4 * oEv] = make_uint4(ctx, fc, 0xFFFFFFFFu, cell);
                    packStore(Eo + (size_t)EV_U4 * oEv + 1, 1, p2);
                    oEv++;
                }
            }
        }
    }
    // per-block counters (summed by k_mvs_finish): one writer per slot and launch, launches are ordered
    unsigned long long v[4] = {cChildren, cExp, cChildren, cRows}; // (every child gets its matrix row)
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[j] += __shfl_xor(v[j], d);
    }
    __shared__ unsigned long long shc[4][4];
    if ((threadIdx.x & 63u) == 0)
        for (int j = 0; j < 4; j++) shc[threadIdx.x >> 6][j] = v[j];
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long t = shc[0][threadIdx.x] + shc[1][threadIdx.x] + shc[2][threadIdx.x] + shc[3][threadIdx.x];
        if (t) B.blockCnt[(size_t)bid * 4 + threadIdx.x] += t;
    }
    if (flags) atomicOr(&q.cnt[3], flags);
}

template <class Geo = GeoN>
__global__ void __launch_bounds__(256)
k_mvs_start(const DevStrategyKT<Geo::MP>* __restrict__ stp, MvBufs B, const MvTask* __restrict__ tasks, uint32_t nTasks, const uint64_t* __restrict__ offs,
            uint32_t gw, const uint32_t* __restrict__ G, const PartOutT<Geo::MP>* __restrict__ parts, Queues q) {
    if (blockStopped(q)) return;
    bfsHeavy<true, MvTraits, Geo>(stp, B, 0u, tasks, nTasks, offs, gw, G, parts, q, blockIdx.x, gridDim.x);
}
constexpr uint32_t MVS_CHAIN = 3; // expansions a lane makes in a row while each yields exactly one plain node (CMB_MVS_CHAIN)
// Round 4: k_mvs_pass ran at 246 registers = 2 wavefronts per SIMD.  With the match words of the row block in LDS, the band geometry kept
// packed, 32-bit counters and the children packed one by one as the scan's results are turned into ranges (moveChildrenCounted<true>: the
// four unpacked pairs never exist side by side) the common instances need 175, and fit 168 = 3 wavefronts per SIMD with four spilled
// words outside the tile loop.  A grid that is resident in ONE round then pays (640 expanding + 128 event blocks = 3 per CU): BASELINE
// configs[4] stand-in 942 k -> 1 123 k reads/s (2 wavefronts: 942 k at 896 + 128 blocks, 853 k at 384 + 128; 4 wavefronts = 128 registers
// spill 50 words inside the loop: 770 k).  The wide geometries (8 ... 13 errors) stay at 2.
#ifndef CMB_MVS_WAVES
#define CMB_MVS_WAVES 3 // wavefronts per SIMD the register allocation of k_mvs_pass<GeoN32 / GeoN> is held to
#endif
// ... and of its instances on 32-bit positions with the children in slots (mvExpandSlots: 154 registers; at 128 two dozen words are
// spilled at the start of a tile and reloaded in its output stage, none inside the chain loop): four wavefronts per SIMD, 896 + 128
// blocks resident in one round, chains of 4: 1 153 k -> 1 209 k reads/s on the configs[4] stand-in
#ifndef CMB_MVS_WAVES_SMALL
#define CMB_MVS_WAVES_SMALL 4
#endif
constexpr uint32_t MVS_GRID_X = 640, MVS_GRID_X_WIDE = 896, MVS_GRID_X_SMALL = 896; // expanding blocks of k_mvs_pass (CMB_MVS_GRID); + BFS_GRID_EV event blocks
constexpr uint32_t MVS_CHAIN_SMALL = 4;
// SMALL: text and run counts below 2^32 — the expanding half works on 32-bit positions (the reference's default build of length_t; the
// records keep their 40-bit fields, so the event half and every other kernel of the backend are the same)
template <class Geo = GeoN, bool SMALL = false>
__global__ void __launch_bounds__(256, Geo::MP == MAXP ? (SMALL ? CMB_MVS_WAVES_SMALL : CMB_MVS_WAVES) : 2)
k_mvs_pass(MoveDev ix, const DevStrategyKT<Geo::MP>* __restrict__ stp, MvBufs B, uint32_t pass, const uint64_t* __restrict__ offs, uint32_t gw,
           const uint32_t* __restrict__ G, const PartOutT<Geo::MP>* __restrict__ parts, Queues q) {
    if (blockStopped(q)) return;
    if (blockIdx.x < B.gridX) {
        if (SMALL) mvExpandSlots<Geo>(ix, B, pass, q, blockIdx.x, B.gridX);
        else mvExpand<Geo, uint64_t>(ix, B, pass, q, blockIdx.x, B.gridX);
    }
    else bfsHeavy<false, MvTraits, Geo>(stp, B, pass, nullptr, 0u, offs, gw, G, parts, q, blockIdx.x - B.gridX, B.gridEv);
}
__global__ void k_mvs_finish(MvBufs B, Queues q) { // one block: per-block counters -> the batch counters
    __shared__ unsigned long long s[4];
    if (threadIdx.x < 4) s[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < BFS_GRID * 4; j += blockDim.x) {
        const unsigned long long v = B.blockCnt[j];
        if (v) atomicAdd(&s[j & 3u], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&q.counters[0], s[0]);  // NODE_COUNTER
        atomicAdd(&q.counters[7], s[1]);  // EXPANSIONS
        atomicAdd(&q.counters[12], s[1]); // DFS_EXPANSIONS
        atomicAdd(&q.counters[11], s[2]); // MATRIX_ROWS
        atomicAdd(&q.counters[13], s[3]); // table rows fetched
        atomicAdd(&q.counters[14], s[3]); // ... by the frontier search
    }
}

// ------------------------------------------------------------------ Hamming distance: the frontier without a matrix
// IndexInterface::recApproxMatchHamming (indexinterface.cpp:1211-1304, RUN_LENGTH_COMPRESSION branches: no in-text switch).  As
// dev_bfs_hamming.hpp on this backend: a node is the range pair (3 planes) + one plane {rsId, scheme | search << 4 | idx << 9 |
// row << 13, mismatches | startDepth << 8, pb | pe << 9}; a child that completes its part enters the next phase at once; a child
// that completes the last part is an in-index occurrence.  One row per pass.
struct MvHbfsBufs {
    uint4* Q[2];
    uint32_t qCap;
    uint32_t* nq;                 // [pass]
    unsigned long long* blockCnt; // [BFS_GRID][4]: children, expansions, -, table rows
    MvFmRec* fmX;
};
template <bool START, int MP = MAXP>
__global__ void __launch_bounds__(256)
k_mvs_hbfs(MoveDev ix, const DevStrategyKT<MP>* __restrict__ stp, MvHbfsBufs B, uint32_t pass, const MvTask* __restrict__ tasks, uint32_t nTasks,
           uint32_t maxLen, const uint8_t* __restrict__ seq, const PartOutT<MP>* __restrict__ parts, Queues q) {
    __shared__ uint32_t sh[4][5];
    if (blockStopped(q)) return;
    constexpr uint32_t PU = MvTraits::PAIR_U4;
    const DevStrategyKT<MP>& st = *stp;
    const uint32_t outP = START ? 0u : pass + 1u;
    const uint32_t nIn = START ? nTasks : min(B.nq[pass], B.qCap);
    const uint4* __restrict__ Qi = B.Q[pass & 1u];
    uint4* __restrict__ Qo = B.Q[outP & 1u];
    const uint32_t qCap = B.qCap;
    unsigned long long cNode = 0, cExp = 0, cRows = 0;
    uint32_t flags = 0;
    for (uint32_t base = blockIdx.x * 256u; base < nIn; base += gridDim.x * 256u) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        uint32_t kinds = 0; // 4 bits per child: 1 node, 3 in-index occurrence
        MvPair ch[4];
        uint32_t cMeta[4], cVd[4], cPw[4];
        uint32_t rsId = 0, fmDepth = 0, nNode = 0, nFm = 0;
        if (START) {
            if (i < nIn) {
                const MvTask t = tasks[i];
                const DevSearchT<MP>& s = st.sch[t.scheme].s[t.search];
                const PartOutT<MP> po = parts[t.rsId];
                rsId = t.rsId;
                ch[0] = loadPair(t.r);
                cMeta[0] = (uint32_t)t.scheme | ((uint32_t)t.search << 4) | ((uint32_t)t.idx << 9);
                cVd[0] = t.depth << 8;
                cPw[0] = (uint32_t)po.pb[s.order[t.idx]] | ((uint32_t)po.pe[s.order[t.idx]] << 9);
                kinds = 1;
                nNode = 1;
            }
        } else if (i < nIn) {
            const uint4 n1 = Qi[(size_t)PU * qCap + i];
            const MvPair parent = MvTraits::load(Qi + i, qCap);
            rsId = n1.x;
            const uint32_t scheme = n1.y & 15u, search = (n1.y >> 4) & 31u, idx = (n1.y >> 9) & 15u, row = n1.y >> 13;
            const uint32_t v = n1.z & 0xFFu, smDepth = n1.z >> 8;
            const uint32_t pb = n1.w & 0x1FFu, pe = (n1.w >> 9) & 0x1FFu;
            const DevSearchT<MP>& s = st.sch[scheme].s[search];
            const uint32_t dir = s.dir[idx];
            const int md = (s.uniAll || idx >= (uint32_t)s.uniIdx) ? 2 : (dir == 0 ? 0 : 1);
            const uint32_t xLen = pe - pb;
            const uint32_t pc = seq[(size_t)rsId * maxLen + (dir == 0 ? pb + row : pe - row - 1)];
            uint32_t rows = 0;
            const uint32_t mask = moveChildrenCounted(ix, md, parent, ch, rows);
            cRows += rows;
            cExp++;
            const uint32_t row1 = row + 1;
            fmDepth = smDepth + xLen;
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                if (!(mask >> c & 1u)) continue;
                cNode++;
                const uint32_t v1 = v + (c + 1 != pc ? 1u : 0u);
                if (v1 > s.U[idx]) continue; // backtrack
                if (row1 == xLen) {          // end of the part
                    if (v1 < s.L[idx]) continue;
                    if (idx == (uint32_t)s.n - 1) {
                        kinds |= 3u << (4 * c);
                        cVd[c] = v1;
                        nFm++;
                    } else { // recApproxMatchHamming(s, match, ..., idx + 1): the child is the start match
                        const PartOutT<MP> po = parts[rsId];
                        kinds |= 1u << (4 * c);
                        cMeta[c] = scheme | (search << 4) | ((idx + 1) << 9);
                        cVd[c] = v1 | (fmDepth << 8);
                        cPw[c] = (uint32_t)po.pb[s.order[idx + 1]] | ((uint32_t)po.pe[s.order[idx + 1]] << 9);
                        nNode++;
                    }
                    continue;
                }
                kinds |= 1u << (4 * c);
                cMeta[c] = scheme | (search << 4) | (idx << 9) | (row1 << 13);
                cVd[c] = v1 | (smDepth << 8);
                cPw[c] = n1.w;
                nNode++;
            }
        }
        const uint32_t want[4] = {nNode, 0u, nFm, 0u};
        uint32_t got[4];
        blockAppend4(&B.nq[outP], &q.cnt[0], &q.cnt[1], &q.cnt[1], want, sh, got);
        uint32_t oNode = got[0], oFm = got[2];
        bool ok = true;
        if (oNode + nNode > qCap) { ok = false; flags |= FLAG_BFS_Q; }
        if (oFm + nFm > q.fmCap) { ok = false; flags |= FLAG_FMOCC_OVERFLOW; }
        if (ok) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t kd = (kinds >> (4 * c)) & 15u;
                if (kd == 1) {
                    MvTraits::store(Qo + oNode, qCap, ch[c]);
                    Qo[(size_t)PU * qCap + oNode] = make_uint4(rsId, cMeta[c], cVd[c], cPw[c]);
                    oNode++;
                } else if (kd == 3) {
                    MvFmRec f;
                    f.rsId = rsId, f.depth = fmDepth, f.dist = cVd[c], f.shift = 0;
                    f.r = storePair(ch[c]);
                    B.fmX[oFm++] = f;
                }
            }
        }
    }
    unsigned long long v2[4] = {cNode, cExp, 0ull, cRows};
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v2[j] += __shfl_xor(v2[j], d);
    }
    __shared__ unsigned long long shc[4][4];
    if ((threadIdx.x & 63u) == 0)
        for (int j = 0; j < 4; j++) shc[threadIdx.x >> 6][j] = v2[j];
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long t = shc[0][threadIdx.x] + shc[1][threadIdx.x] + shc[2][threadIdx.x] + shc[3][threadIdx.x];
        if (t) B.blockCnt[(size_t)blockIdx.x * 4 + threadIdx.x] += t;
    }
    if (flags) atomicOr(&q.cnt[3], flags);
}

// ------------------------------------------------------------------ naive backtracking
// IndexInterface::approxMatchesNaive / approxMatchesNaiveHamming (indexinterface.cpp:1055-1209, RUN_LENGTH_COMPRESSION branches: no
// in-text switch) for the reads k_mvs_parts marks (psel bit 7: not longer than the number of parts, or a one-part strategy —
// searchstrategy.cpp:148-152, :442-459).  As dev_bfs_naive.hpp on this backend: a node is the range pair (3 planes) + one plane
// {rsId, row | score << 16, RAC bit | mismatches << 8, -} + (edit distance) one plane {HP, HN}; the whole pattern is matched backward
// from the complete range (unidirectional: mode 2), every final-column node within the bound is an in-index occurrence.
constexpr uint32_t MVS_NAIVE_STOP = FLAG_NAIVE_Q | FLAG_FMOCC_OVERFLOW;
template <bool EDIT, bool START, bool NARROW = false /* the matrix of 11 ... 13 errors: 16-row blocks (dev_matrix.hpp: MXN_*) */>
__global__ void __launch_bounds__(256)
k_mvs_naive(MoveDev ix, MvHbfsBufs B, uint32_t pass, const uint8_t* __restrict__ psel, uint32_t tasksRS, const uint64_t* __restrict__ offs,
            uint32_t gw, const uint32_t* __restrict__ G, const uint8_t* __restrict__ seq, uint32_t maxLen, uint32_t k, Queues q) {
    using Mx = typename std::conditional<NARROW, MxNarrow, MxRef64>::type;
    __shared__ uint32_t sh[4][5];
    __shared__ uint32_t stopWord;
    if (threadIdx.x == 0) stopWord = __hip_atomic_load(&q.cnt[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & MVS_NAIVE_STOP;
    __syncthreads();
    if (stopWord) return; // (an earlier pass dropped nodes: the host grows the queue and runs the search again)
    constexpr uint32_t PU = MvTraits::PAIR_U4;
    const uint32_t outP = START ? 0u : pass + 1u;
    const uint32_t nIn = START ? tasksRS : min(B.nq[pass], B.qCap);
    const uint4* __restrict__ Qi = B.Q[pass & 1u];
    uint4* __restrict__ Qo = B.Q[outP & 1u];
    const uint32_t qCap = B.qCap;
    uint32_t cNode = 0, cExp = 0, cMx = 0, cRows = 0, flags = 0;
    for (uint32_t base = blockIdx.x * 256u; base < nIn; base += gridDim.x * 256u) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        uint32_t kinds = 0; // per child: bit 0 node, bit 2 in-index occurrence
        MvPair ch[4];
        uint64_t cHP[4], cHN[4];
        uint32_t cState[4], cAux[4], cDist[4];
        uint32_t rsId = 0, row1 = 0, nNode = 0, nFm = 0;
        if (START) {
            if (i < nIn && (psel[i] & 0x80u) && !(!EDIT && offs[(i >> 1) + 1] == offs[i >> 1])) { // (Hamming, empty read: pattern[-1] in the reference)
                rsId = i;
                ch[0] = mvCompleteRange(ix);
                const uint64_t HP0 = (~0ull) << Mx::LEFT;
                cHP[0] = HP0;
                cHN[0] = ~HP0;
                cState[0] = 0;
                cAux[0] = Mx::DIAG + k;
                kinds = 1;
                nNode = 1;
            }
        } else if (i < nIn) {
            const uint4 n1 = Qi[(size_t)PU * qCap + i];
            const MvPair parent = MvTraits::load(Qi + i, qCap);
            rsId = n1.x;
            const uint32_t row = n1.y & 0xFFFFu, score = n1.y >> 16, rac = n1.z & 0xFFu, v = n1.z >> 8;
            const uint32_t len = (uint32_t)(offs[(rsId >> 1) + 1] - offs[rsId >> 1]);
            uint64_t pHP = 0, pHN = 0;
            MatGeom g;
            g.n = len + 1;
            g.maxED = k;
            g.Wv = k;
            g.Wh = k;
            g.m = max(g.Wv + g.n, g.Wv + g.Wh + 1u);
            if (EDIT) {
                const uint4 n2 = Qi[(size_t)(PU + 1) * qCap + i];
                pHP = u64of(n2.x, n2.y);
                pHN = u64of(n2.z, n2.w);
            }
            uint32_t rows = 0;
            const uint32_t mask = moveChildrenCounted(ix, 2, parent, ch, rows);
            cRows += rows;
            cExp++;
            row1 = row + 1;
            const uint32_t pc = EDIT ? 0u : seq[(size_t)rsId * maxLen + (len - row1)];
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                if (!(mask >> c & 1u)) continue;
                cNode++;
                uint32_t kd = 0;
                if (EDIT) {
                    if (row1 >= g.m) continue; // (:1098)
                    cMx++;
                    uint64_t HP = pHP, HN = pHN, D0, RAC = 1ull << rac;
                    uint32_t sc = score;
                    const uint64_t M = matchWord<Mx::LEFT, Mx::BLOCK>(gString(G, gw, rsId, 1u, c), 0u, len, row1 / Mx::BLOCK);
                    if (!Mx::row(g, row1, M, HP, HN, D0, RAC, sc)) continue;
                    if (g.inFinalColumn(row1)) {
                        const uint32_t d = Mx::cell(row1, len, HP, HN, sc);
                        if (d <= k) {
                            kd |= 4u;
                            cDist[c] = d;
                        }
                    }
                    kd |= 1u; // (no in-text verification on this index: every valid node is extended)
                    cHP[c] = HP;
                    cHN[c] = HN;
                    cState[c] = row1 | (sc << 16);
                    cAux[c] = (uint32_t)__ffsll((unsigned long long)RAC) - 1u;
                } else {
                    const uint32_t v1 = v + (c + 1 != pc ? 1u : 0u);
                    if (v1 > k) continue;
                    if (row1 == len) {
                        kd = 4u;
                        cDist[c] = v1;
                    } else {
                        kd = 1u;
                        cState[c] = row1;
                        cAux[c] = v1 << 8;
                    }
                }
                kinds |= kd << (4 * c);
                nNode += kd & 1u;
                nFm += (kd >> 2) & 1u;
            }
        }
        const uint32_t want[4] = {nNode, 0u, nFm, 0u};
        uint32_t got[4];
        blockAppend4(&B.nq[outP], &q.cnt[0], &q.cnt[1], &q.cnt[1], want, sh, got);
        uint32_t oNode = got[0], oFm = got[2];
        bool ok = true;
        if (oNode + nNode > qCap) { ok = false; flags |= FLAG_NAIVE_Q; }
        if (oFm + nFm > q.fmCap) { ok = false; flags |= FLAG_FMOCC_OVERFLOW; }
        if (ok && kinds) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t kd = (kinds >> (4 * c)) & 15u;
                if (kd & 4u) {
                    MvFmRec f;
                    f.rsId = rsId, f.depth = row1, f.dist = cDist[c], f.shift = 0;
                    f.r = storePair(ch[c]);
                    B.fmX[oFm++] = f;
                }
                if (kd & 1u) {
                    MvTraits::store(Qo + oNode, qCap, ch[c]);
                    Qo[(size_t)PU * qCap + oNode] = make_uint4(rsId, cState[c], cAux[c], 0u);
                    if (EDIT) Qo[(size_t)(PU + 1) * qCap + oNode] = make_uint4((uint32_t)cHP[c], (uint32_t)(cHP[c] >> 32), (uint32_t)cHN[c], (uint32_t)(cHN[c] >> 32));
                    oNode++;
                }
            }
        }
    }
    const uint32_t local[4] = {cNode, cExp, cMx, cRows};
    const int which[4] = {0, 7, 11, 13}; // NODE_COUNTER, EXPANSIONS, MATRIX_ROWS, table rows fetched
    flushCounters(q, local, which, 4);
    if (flags) atomicOr(&q.cnt[3], flags);
}

// ------------------------------------------------------------------ in-index occurrences -> text occurrences
// Occurrences::eraseDoublesFM (indexhelpers.h:2135-2146) under the RLC flavour's equality (FMOcc::== over SARangePair::==,
// which includes run indices, toehold, toeholdRepresentsEnd and originalDepth, :1226-1233): records are sorted by
// read x strand | a hash of everything else, and a record is dropped if it equals its predecessor field by field.
__device__ __forceinline__ uint64_t mix64(uint64_t h, uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xFF51AFD7ED558CCDull;
    return h ^ (h >> 33);
}
__device__ __forceinline__ bool sameFm(const MvFmRec& a, const MvFmRec& b) {
    return a.rsId == b.rsId && a.depth == b.depth && a.dist == b.dist && a.shift == b.shift && a.r.begin == b.r.begin && a.r.end == b.r.end &&
           a.r.beginRun == b.r.beginRun && a.r.endRun == b.r.endRun && a.r.toehold == b.r.toehold && a.r.repEnd == b.r.repEnd &&
           a.r.depth == b.r.depth;
}
__global__ void k_mvs_fm_keys(const MvFmRec* __restrict__ fm, uint32_t n, unsigned long long* __restrict__ keys, uint32_t* __restrict__ idx) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const MvFmRec f = fm[i];
        idx[i] = i;
        if (f.rsId == 0xFFFFFFFFu) { // holes go last
            keys[i] = ~0ull;
            continue;
        }
        uint64_t h = mix64(0, f.r.begin);
        h = mix64(h, f.r.end);
        h = mix64(h, f.r.beginRun);
        h = mix64(h, f.r.endRun);
        h = mix64(h, f.r.toehold);
        h = mix64(h, ((uint64_t)f.depth << 32) | f.r.depth);
        h = mix64(h, ((uint64_t)f.dist << 32) | ((uint64_t)f.shift << 1) | f.r.repEnd);
        keys[i] = ((uint64_t)f.rsId << 38) | (h >> 26);
    }
}
// keep[i] = 1 for the first record of every run of equal records (holes: 0); widths[i] = its SA range width
__global__ void k_mvs_fm_unique(const MvFmRec* __restrict__ fm, const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ keep,
                                uint64_t* __restrict__ widths) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const MvFmRec f = fm[idx[i]];
        bool k = f.rsId != 0xFFFFFFFFu;
        if (k && i > 0) k = !sameFm(f, fm[idx[i - 1]]);
        keep[i] = k ? 1u : 0u;
        widths[i] = k ? f.r.end - f.r.begin : 0ull;
    }
}
// compact the kept records (slot[i] = exclusive scan of keep) into ranges for k_move_locate + their meta
__global__ void k_mvs_fm_compact(const MvFmRec* __restrict__ fm, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ keep,
                                 const uint32_t* __restrict__ slot, uint32_t n, MoveRangeRec* __restrict__ ranges, uint4* __restrict__ meta,
                                 uint64_t* __restrict__ widthsOut) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!keep[i]) continue;
        const MvFmRec f = fm[idx[i]];
        const uint32_t o = slot[i];
        ranges[o] = f.r;
        meta[o] = make_uint4(f.rsId, f.depth, f.dist, f.shift);
        widthsOut[o] = f.r.end - f.r.begin;
    }
}
// one lane per located position: TextOcc = [pos + shift, pos + shift + depth) (indexinterface.cpp:1410-1416) as a sort key
// (read | begin) and a value (distance, width, strand) that orders equal begins as TextOcc::operator< does (:779-795)
// which: 0 every record, groups = reads; 1 only the reads matched by naive backtracking (psel bit 7), groups = read x strand — that
// path filters its occurrences per strand first (indexinterface.cpp:1137, :1205); 2 all other reads (the survivors of pass 1 join
// them through k_mvs_occ_keys).  Records left out get the all-ones key, which sorts behind every group.
__global__ void k_mvs_text_keys(const uint64_t* __restrict__ positions, const uint64_t* __restrict__ recOff, uint32_t nRecs, uint64_t total,
                                const uint4* __restrict__ meta, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals,
                                uint32_t* __restrict__ bad, const uint8_t* __restrict__ psel = nullptr, int which = 0, int perStrand = 0) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < total; j += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = nRecs; // the last record whose offset is <= j
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (recOff[mid] <= j) lo = mid;
            else hi = mid;
        }
        const uint4 m = meta[lo];
        const uint64_t begin = positions[j] + m.w;
        if (begin >> 40 || m.x >> 24 || m.y >= (1u << 19) || m.z >= 16u) atomicAdd(bad, 1u);
        const bool naive = which != 0 && (psel[m.x] & 0x80u) != 0u;
        const uint64_t grp = (which == 1 || perStrand) ? m.x : m.x >> 1; // (perStrand: every strand filtered by itself, BEST mode's mapRead)
        const bool leftOut = (which == 1 && !naive) || (which == 2 && naive);
        keys[j] = leftOut ? ~0ull : ((grp << 40) | (begin & ((1ull << 40) - 1)));
        vals[j] = (m.z << 20) | (m.y << 1) | (m.x & 1u);
    }
}
// The redundancy filter of getUniqueTextOccurrences (indexinterface.cpp:1445-1485), one lane per read over its sorted segment.
// Among the occurrences of one begin position only the smallest (distance, width) can ever be kept (the others are skipped
// as `diff == 0` if it is kept, and fail the same test it failed if it is not), so each group of equal begins is reduced to
// its minimum first; strand 0 wins a tie, as in the FM-index backend.  WRITE = false counts, true writes.
struct MoveOccOut {
    uint64_t begin, end;
    uint32_t distance, strand;
};
// uniqueOnly: getTextOccHamming (indexinterface.cpp:1331-1371) — sort + unique under TextOcc::operator== (range and distance): per
// begin every distinct (distance, width) is kept, in ascending order, strand 0 winning a tie (the two strands of a read can match
// the same range at different distances).
template <bool WRITE>
__global__ void k_mvs_filter(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t total, uint32_t nReads,
                             uint32_t maxED, uint64_t* __restrict__ counts, const uint64_t* __restrict__ outOff, MoveOccOut* __restrict__ out,
                             uint32_t uniqueOnly) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nReads; r += gridDim.x * blockDim.x) {
        const uint64_t kLo = (uint64_t)r << 40, kHi = ((uint64_t)r + 1) << 40;
        uint64_t lo = 0, hi = total; // first key >= kLo
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (keys[mid] < kLo) lo = mid + 1;
            else hi = mid;
        }
        uint64_t j = lo, nKept = 0;
        if (uniqueOnly) {
            uint64_t w = WRITE ? outOff[r] : 0;
            while (j < total && keys[j] < kHi) {
                const uint64_t key = keys[j], begin = key & ((1ull << 40) - 1);
                uint64_t e = j + 1;
                while (e < total && keys[e] == key) e++;
                uint32_t lastV = 0;
                bool first = true;
                for (;;) { // the group's values in ascending order, one per distinct (distance, width)
                    uint32_t best = 0xFFFFFFFFu;
                    for (uint64_t t = j; t < e; t++) {
                        const uint32_t v = vals[t];
                        if ((first || (v >> 1) > (lastV >> 1)) && v < best) best = v;
                    }
                    if (best == 0xFFFFFFFFu) break;
                    if (WRITE) out[w++] = MoveOccOut{begin, begin + ((best >> 1) & 0x7FFFFu), best >> 20, best & 1u};
                    nKept++;
                    lastV = best;
                    first = false;
                }
                j = e;
            }
            if (!WRITE) counts[r] = nKept;
            continue;
        }
        const uint64_t maxDiff = 2ull * maxED;
        uint64_t prevBegin = ~0ull;
        uint32_t prevED = maxED + 1, prevDepth = 0xFFFFFFFFu;
        MoveOccOut last{};
        bool have = false;
        uint64_t w = WRITE ? outOff[r] : 0;
        while (j < total && keys[j] < kHi) {
            const uint64_t begin = keys[j] & ((1ull << 40) - 1);
            uint32_t best = vals[j];
            j++;
            while (j < total && keys[j] == (kLo | begin)) {
                best = min(best, vals[j]);
                j++;
            }
            const uint32_t dist = best >> 20, width = (best >> 1) & 0x7FFFFu, strand = best & 1u;
            const uint64_t diff = begin > prevBegin ? begin - prevBegin : prevBegin - begin;
            if (diff == 0) continue;
            if (diff <= maxDiff) {
                if (dist > prevED || (dist == prevED && width >= prevDepth)) continue;
                nKept--; // the previous one was worse: pop_back
                have = false;
            }
            if (have) {
                if (WRITE) out[w++] = last;
                have = false;
            }
            prevBegin = begin, prevED = dist, prevDepth = width;
            last = MoveOccOut{begin, begin + width, dist, strand};
            have = true;
            nKept++;
        }
        if (have && WRITE) out[w++] = last;
        if (!WRITE) counts[r] = nKept;
    }
}

// the survivors of the naive path's own filter pass (groups = read x strand) as keys of their READ, behind the keys of the other
// reads: they pass the filter of the mapping mode a second time (searchstrategy.cpp:455-457)
__global__ void k_mvs_occ_keys(const MoveOccOut* __restrict__ occ, const uint64_t* __restrict__ occOff, uint32_t nGroups,
                               unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals, int perStrand = 0) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nGroups; g += gridDim.x * blockDim.x)
        for (uint64_t e = occOff[g]; e < occOff[g + 1]; e++) {
            const MoveOccOut o = occ[e];
            keys[e] = ((uint64_t)(perStrand ? g : g >> 1) << 40) | o.begin;
            vals[e] = (o.distance << 20) | ((uint32_t)(o.end - o.begin) << 1) | (g & 1u);
        }
}

// offsets per read from offsets per read x strand (the per-strand filter of BEST mode: the two strands of a read are neighbours)
__global__ void k_mvs_read_offsets(const uint64_t* __restrict__ rsOff, uint32_t nReads, uint64_t* __restrict__ readOff) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= nReads; r += gridDim.x * blockDim.x) readOff[r] = rsOff[2 * (size_t)r];
}

// the final occurrences as k_cigar takes them: {begin, end, distance, strand} in 32 bits (texts below 2^32) and the read of each
__global__ void k_mvs_occ32(const MoveOccOut* __restrict__ occ, const uint64_t* __restrict__ occOff, uint32_t nReads, uint4* __restrict__ out,
                            uint32_t* __restrict__ outRead) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nReads; r += gridDim.x * blockDim.x)
        for (uint64_t e = occOff[r]; e < occOff[r + 1]; e++) {
            const MoveOccOut o = occ[e];
            out[e] = make_uint4((uint32_t)o.begin, (uint32_t)o.end, o.distance, o.strand);
            outRead[e] = r;
        }
}

} // namespace cmb

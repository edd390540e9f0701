// Device side of the run-length compressed backend (b-move): the move table in HBM, run walks,
// LF with fast-forward, character extension with toehold maintenance, locate by phi / phi^-1 under the PLCP bound.
// gfx950 only.
//
// Reference (paths relative to /root/reference/src): bmove/moverepr.{h,cpp} (MoveLFReprBP), bmove/bmove.{h,cpp}
// (BMove), bmove/plcp.h, bmove/sparsebitvec.h; range types indexhelpers.h:137-255, :1040-1260.
//
// Layout.  The reference packs a row into ceil((3 + 2*ceil(log2 n) + ceil(log2 r)) / 8) bytes and reads every field
// through an unaligned 128-bit load (moverepr.cpp:36-47).  Here a row is ONE aligned 16-byte word:
//     bits   0..2   run head (0 = '$', 1..4 = ACGT)
//     bits   3..42  inputStartPos   (40 bits: texts up to 2^40 - 1 characters)
//     bits  43..82  outputStartPos  (40 bits)
//     bits  83..122 outputStartRun  (40 bits)
//     bits 123..127 gap = min(31, inputStartPos[outputStartRun + 1] - outputStartPos): how far the LF image of the run's
//                    first position is from the next run boundary.  An LF image out + offset with offset < gap lies in
//                    outputStartRun itself — the fast-forward (moverepr.cpp:283-293) then needs no table access at all, and
//                    with gap < 31 <= ... otherwise it starts one run further on.  (k_move_gaps fills it after the unpacking.)
// so a run walk is a stream of consecutive 16-byte loads (eight rows per 128-byte line), and the whole row of a run
// arrives in one request.  `k_move_unpack` converts the reference's file rows on the device.
//
// Extension.  The reference extends with one character at a time: walkToNextRun + walkToPreviousRun for the range of
// the child, the same two walks again for every smaller character (getCumulativeCounts -> countChar), and a third
// backward walk for the toehold (bmove.cpp:222-266, :328-442).  All of it is a function of, per character, the first and
// the last run of the parent range that holds it.  `moveScan` finds those for all four characters in ONE pass from
// both ends of the range (it stops as soon as every character has been seen), and `moveChildren` derives the four
// children from them: countChar(range, c) is the width of child c, the cumulative count is a prefix sum over the
// children, the toehold run is the last run found for c.  Results are bit-identical to the reference's, field by field.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmb {

constexpr uint64_t MV_M40 = (1ull << 40) - 1;

struct MoveRow {
    uint32_t head;
    uint64_t in, out, outRun;
};

__host__ __device__ inline uint4 packMoveRow(uint32_t head, uint64_t in, uint64_t out, uint64_t outRun) {
    const uint64_t lo = (uint64_t)(head & 7u) | (in & MV_M40) << 3 | (out & MV_M40) << 43;
    const uint64_t hi = (out & MV_M40) >> 21 | (outRun & MV_M40) << 19;
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
__host__ __device__ inline MoveRow unpackMoveRow(const uint4 v) {
    const uint64_t lo = (uint64_t)v.x | (uint64_t)v.y << 32, hi = (uint64_t)v.z | (uint64_t)v.w << 32;
    MoveRow r;
    r.head = (uint32_t)lo & 7u;
    r.in = (lo >> 3) & MV_M40;
    r.out = (lo >> 43 | hi << 21) & MV_M40;
    r.outRun = (hi >> 19) & MV_M40;
    return r;
}
constexpr uint32_t MV_GAP_MAX = 31;
__device__ inline uint32_t rowGap(const uint4 v) { return v.w >> 27; }
__device__ inline uint32_t rowHead(const uint4 v) { return v.x & 7u; }
__device__ inline uint64_t rowIn(const uint4 v) { return (((uint64_t)v.x | (uint64_t)v.y << 32) >> 3) & MV_M40; }

// one direction of the index: rows[0 .. runs] (row `runs` is the terminating row: head 0, inputStartPos = n)
struct MoveTable {
    const uint4* rows;
    uint64_t runs;
    uint64_t zeroCharPos;
    const uint64_t* samplesFirst; // SA value at the first position of every run (buildindex.cpp:942-953)
    const uint64_t* samplesLast;
};

// a sorted set of text positions with a bucket directory in front of it: dir[b] = first element >= b << shift
struct PosSet {
    const uint64_t* pos;
    uint64_t count;
    const uint64_t* dir;
    uint32_t shift;
    // number of elements < x (SparseBitvec::rank, sparsebitvec.h:100-102)
    __device__ uint64_t rank(uint64_t x) const {
        const uint64_t b = x >> shift;
        uint64_t lo = dir[b], hi = dir[b + 1];
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (pos[mid] < x) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    }
};

struct MoveDev {
    uint64_t n; // text length including '$'
    MoveTable fwd, rev;
    // locate
    PosSet predFirst, predLast;       // marked positions of predFirst / predLast (buildindex.cpp:990-1013)
    const uint64_t* firstToRun;       // buildindex.cpp:1044-1066
    const uint64_t* lastToRun;
    PosSet plcpPos;                   // positions where a PLCP run (PLCP[i] + i constant) starts
    const uint64_t* plcpSum;          // PLCP[q] + q at those positions
};

// cmb_move_range (include/columba_amd.h) as the kernels see it
struct MoveRangeRec {
    uint64_t begin, end, beginRun, endRun;
    uint64_t rBegin, rEnd, rBeginRun, rEndRun;
    uint64_t toehold;
    uint32_t depth;
    uint8_t valid, rValid, repEnd, reserved;
};
static_assert(sizeof(MoveRangeRec) == 80, "record layout");

// (P: the type of a text position / run number — 64 bits in general; 32 bits for the frontier kernel's instance on indexes whose text and
// run counts stay below 2^32, the reference's default build of length_t: move_search.hpp, mvExpand)
template <typename P>
struct MvRangeT {
    P begin, end, beginRun, endRun;
    bool valid;
};
typedef MvRangeT<uint64_t> MvRange;
// the fields of a row as positions of type P (P = uint32_t: the low 32 bits of the 40-bit fields, three funnel shifts)
template <typename P> struct MoveRowT {
    uint32_t head;
    P in, out, outRun;
};
template <typename P> __device__ __forceinline__ P rowInT(const uint4 v);
template <> __device__ __forceinline__ uint64_t rowInT<uint64_t>(const uint4 v) { return (((uint64_t)v.x | (uint64_t)v.y << 32) >> 3) & MV_M40; }
template <> __device__ __forceinline__ uint32_t rowInT<uint32_t>(const uint4 v) { return __funnelshift_r(v.x, v.y, 3u); }
template <typename P> __device__ __forceinline__ P rowInT(const uint2 v);
template <> __device__ __forceinline__ uint64_t rowInT<uint64_t>(const uint2 v) { return (((uint64_t)v.x | (uint64_t)v.y << 32) >> 3) & MV_M40; }
template <> __device__ __forceinline__ uint32_t rowInT<uint32_t>(const uint2 v) { return __funnelshift_r(v.x, v.y, 3u); }
template <typename P> __device__ __forceinline__ MoveRowT<P> unpackMoveRowT(const uint4 v);
template <> __device__ __forceinline__ MoveRowT<uint64_t> unpackMoveRowT<uint64_t>(const uint4 v) {
    const MoveRow r = unpackMoveRow(v);
    return MoveRowT<uint64_t>{r.head, r.in, r.out, r.outRun};
}
template <> __device__ __forceinline__ MoveRowT<uint32_t> unpackMoveRowT<uint32_t>(const uint4 v) { // fields at bits 3, 43, 83
    return MoveRowT<uint32_t>{v.x & 7u, __funnelshift_r(v.x, v.y, 3u), __funnelshift_r(v.y, v.z, 11u), __funnelshift_r(v.z, v.w, 19u)};
}

// MoveLFReprBP::getRunIndex / computeRunIndices (moverepr.cpp:213-249): the runs that hold begin and end - 1, searched
// between the (stale but enclosing) run indices the range carries
// (both searches step TOGETHER — two independent probes per memory round trip; the second one runs over the whole interval instead of
// starting at the first one's result, which finds the same run)
template <typename P>
__device__ inline void computeRunIndices(const MoveTable& t, MvRangeT<P>& r) {
    P lo1 = r.beginRun, hi1 = r.endRun, lo2 = r.beginRun, hi2 = r.endRun;
    const P last = r.end - 1;
    while (hi1 > lo1 || hi2 > lo2) {
        const P mid1 = (P)(((uint64_t)lo1 + hi1 + 1) >> 1), mid2 = (P)(((uint64_t)lo2 + hi2 + 1) >> 1);
        const uint2 a = *reinterpret_cast<const uint2*>(t.rows + mid1), b = *reinterpret_cast<const uint2*>(t.rows + mid2);
        const P in1 = rowInT<P>(a), in2 = rowInT<P>(b);
        if (hi1 > lo1) {
            if (in1 <= r.begin) lo1 = mid1;
            else hi1 = mid1 - 1;
        }
        if (hi2 > lo2) {
            if (in2 <= last) lo2 = mid2;
            else hi2 = mid2 - 1;
        }
    }
    r.beginRun = lo1;
    r.endRun = lo2;
    r.valid = true;
}

// per character 1..4: the first and the last position of the range whose run holds it, with that run
struct MoveScan {
    uint32_t found; // bit c set: character c occurs in the range
    uint64_t firstPos[5], firstRun[5], lastPos[5], lastRun[5];
};

// walkToNextRun / walkToPreviousRun (moverepr.cpp:251-281) for all characters at once
__device__ inline void moveScan(const MoveTable& t, const MvRange& r, MoveScan& s) {
    s.found = 0;
    uint64_t run = r.beginRun, pos = r.begin;
    uint4 row = t.rows[run];
    while (true) {
        const uint32_t h = rowHead(row);
        if (!(s.found >> h & 1u)) {
            s.found |= 1u << h;
            s.firstPos[h] = pos;
            s.firstRun[h] = run;
        }
        if ((s.found & 0x1Eu) == 0x1Eu || run == r.endRun) break;
        run++;
        row = t.rows[run];
        pos = rowIn(row);
    }
    uint32_t seen = 0;
    run = r.endRun;
    pos = r.end - 1;
    row = t.rows[run];
    while (true) {
        const uint32_t h = rowHead(row);
        if (!(seen >> h & 1u)) {
            seen |= 1u << h;
            s.lastPos[h] = pos;
            s.lastRun[h] = run;
        }
        // every character of the forward pass is met again at the latest when the walk reaches the run that pass
        // stopped in; characters it did not meet (the forward pass ended early with all four) cannot exist
        if ((seen & 0x1Eu) == (s.found & 0x1Eu) || run == r.beginRun) break;
        pos = rowIn(row) - 1;
        run--;
        row = t.rows[run];
    }
    s.found &= seen | 1u; // (defensive: both passes cover the same set)
}

// MoveLFReprBP::findLF (moverepr.cpp:283-301)
__device__ inline void moveLF(const MoveTable& t, uint64_t& pos, uint64_t& run) {
    const uint4 w = t.rows[run];
    const MoveRow r = unpackMoveRow(w);
    const uint64_t off = pos - r.in;
    const uint32_t gap = rowGap(w);
    pos = r.out + off;
    run = r.outRun;
    if (off < gap) return; // still inside the run the LF image of the run's first position lies in
    if (gap < MV_GAP_MAX) run++;
    while (rowIn(t.rows[run + 1]) <= pos) run++; // fast-forward; the terminating row (inputStartPos = n) stops it
}

template <typename P>
struct MvPairT {
    MvRangeT<P> sa, rev;
    P toehold;
    bool repEnd;
    uint32_t depth;
};
typedef MvPairT<uint64_t> MvPair;

__device__ inline MvPair loadPair(const MoveRangeRec& q) {
    MvPair p;
    p.sa = {q.begin, q.end, q.beginRun, q.endRun, q.valid != 0};
    p.rev = {q.rBegin, q.rEnd, q.rBeginRun, q.rEndRun, q.rValid != 0};
    p.toehold = q.toehold;
    p.repEnd = q.repEnd != 0;
    p.depth = q.depth;
    return p;
}
__device__ inline MoveRangeRec storePair(const MvPair& p) {
    MoveRangeRec q;
    q.begin = p.sa.begin, q.end = p.sa.end, q.beginRun = p.sa.beginRun, q.endRun = p.sa.endRun;
    q.rBegin = p.rev.begin, q.rEnd = p.rev.end, q.rBeginRun = p.rev.beginRun, q.rEndRun = p.rev.endRun;
    q.toehold = p.toehold, q.depth = p.depth;
    q.valid = p.sa.valid, q.rValid = p.rev.valid, q.repEnd = p.repEnd, q.reserved = 0;
    return q;
}

// All four children of a parent: BMove::findRangesWithExtraCharBackward (bmove.cpp:328-382), ...Forward (:384-442),
// ...BackwardUniDirectional (:444-478).  mode 0 forward, 1 backward, 2 uni-directional backward.  Returns the mask of
// non-empty children (bit c - 1).
__device__ inline uint32_t moveChildren(const MoveDev& ix, const int mode, const MvPair& parent, MvPair child[4]) {
    const bool fw = mode == 0;
    const MoveTable& t = fw ? ix.rev : ix.fwd;
    MvRange trivial = fw ? parent.rev : parent.sa;
    const MvRange other = fw ? parent.sa : parent.rev;
    if (!trivial.valid) computeRunIndices(t, trivial); // bmove.cpp:289-297, :394-397
    MoveScan s;
    moveScan(t, trivial, s);
    const uint64_t parentWidth = trivial.end - trivial.begin;
    // MoveLFReprBP::getCumulativeCounts (moverepr.cpp:347-365): the '$' of the range, then the smaller characters
    uint64_t cum = (trivial.begin <= t.zeroCharPos && trivial.end > t.zeroCharPos) ? 1 : 0;
    uint32_t mask = 0;
    for (int c = 1; c <= 4; c++) {
        MvPair& ch = child[c - 1];
        if (!(s.found >> c & 1u)) { // addChar: setEmpty() (moverepr.cpp:313-316), SARangePair(range1, range1, 0, false, 0)
            ch.sa = {0, 0, 0, 0, false};
            ch.rev = ch.sa;
            ch.toehold = 0, ch.repEnd = false, ch.depth = 0;
            continue;
        }
        mask |= 1u << (c - 1);
        uint64_t p1 = s.firstPos[c], r1 = s.firstRun[c], p2 = s.lastPos[c], r2 = s.lastRun[c];
        moveLF(t, p1, r1);
        moveLF(t, p2, r2);
        const MvRange range1 = {p1, p2 + 1, r1, r2, true};
        const uint64_t width = p2 + 1 - p1; // = countChar(trivial, c) (moverepr.cpp:329-345)
        MvRange second;
        if (width == parentWidth) { // the other range and the toehold carry over
            second = mode == 2 ? MvRange{0, 0, 0, 0, true} : other;
            ch.toehold = fw ? parent.toehold + (parent.repEnd ? 1 : 0) : parent.toehold - (parent.repEnd ? 0 : 1);
            ch.repEnd = parent.repEnd;
        } else {
            second = mode == 2 ? MvRange{0, 0, 0, 0, true} : MvRange{other.begin + cum, other.begin + cum + width, other.beginRun, other.endRun, false};
            // BMove::computeToehold / computeToeholdRev (bmove.cpp:222-266): the last run of the range that holds c
            const uint64_t smp = s.lastRun[c] == trivial.endRun ? t.samplesFirst[trivial.endRun] : t.samplesLast[s.lastRun[c]];
            ch.toehold = fw ? ix.n - 1 - (smp - 1) : smp - 1;
            ch.repEnd = fw;
        }
        ch.sa = fw ? second : range1;
        ch.rev = fw ? range1 : second;
        ch.depth = parent.depth + 1;
        cum += width;
    }
    return mask;
}

// PLCP::operator[] (bmove/plcp.h:165-172) on the run-length form: PLCP[i] + i is constant from one sampled position
// to the next
__device__ inline uint64_t plcpAt(const MoveDev& ix, uint64_t i) {
    const uint64_t k = ix.plcpPos.rank(i + 1); // sampled positions <= i; position 0 is always sampled
    return ix.plcpSum[k - 1] - i;
}
// BMove::phi / phiInverse (bmove.cpp:178-217)
__device__ inline uint64_t movePhi(const MoveDev& ix, uint64_t pos) {
    const uint64_t rk = ix.predFirst.rank(pos);
    const uint64_t predRank = rk == 0 ? ix.predFirst.count - 1 : rk - 1;
    const uint64_t pred = ix.predFirst.pos[predRank];
    const uint64_t delta = pred < pos ? pos - pred : pos + 1;
    const uint64_t run = ix.firstToRun[predRank];
    if (run == 0) return ix.n; // phi of the smallest suffix: not a position (the reference asserts, bmove.cpp:189)
    const uint64_t prevSample = ix.fwd.samplesLast[run - 1];
    return (prevSample + delta - 1) % ix.n;
}
__device__ inline uint64_t movePhiInverse(const MoveDev& ix, uint64_t pos) {
    const uint64_t rk = ix.predLast.rank(pos);
    const uint64_t predRank = rk == 0 ? ix.predLast.count - 1 : rk - 1;
    const uint64_t pred = ix.predLast.pos[predRank];
    const uint64_t delta = pred < pos ? pos - pred : pos + 1;
    const uint64_t run = ix.lastToRun[predRank];
    if (run + 1 >= ix.fwd.runs) return ix.n; // phi^-1 of the largest suffix (bmove.cpp:210)
    const uint64_t nextSample = ix.fwd.samplesFirst[run + 1];
    return (nextSample + delta - 1) % ix.n;
}

// ---- kernels ---------------------------------------------------------------------------------------------------------

// rows of a .LFBP file (moverepr.cpp:170-181: head at bit 0, inputStartPos at bit 3, outputStartPos at 3 + bitsN,
// outputStartRun at 3 + 2 bitsN) -> 16-byte rows
__global__ void k_move_unpack(const uint8_t* __restrict__ packed, uint64_t rowsTotal, uint32_t rowBytes, uint32_t bitsN, uint32_t bitsR,
                              uint4* __restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < rowsTotal; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t* p = packed + i * rowBytes;
        uint64_t lo = 0, hi = 0;
        for (uint32_t b = 0; b < rowBytes; b++) {
            if (b < 8) lo |= (uint64_t)p[b] << (8 * b);
            else hi |= (uint64_t)p[b] << (8 * (b - 8));
        }
        auto field = [&](uint32_t off, uint32_t len) -> uint64_t { // len <= 40, off + len <= 128
            const uint64_t v = off >= 64 ? hi >> (off - 64) : off == 0 ? lo : (lo >> off) | (hi << (64 - off));
            return v & ((1ull << len) - 1);
        };
        const uint32_t head = (uint32_t)lo & 7u;
        const uint64_t in = field(3, bitsN), o = field(3 + bitsN, bitsN), orun = field(3 + 2 * bitsN, bitsR);
        out[i] = packMoveRow(head, in, o, orun);
    }
}

// the gap field of every row (see the layout above); after k_move_unpack
__global__ void k_move_gaps(uint4* __restrict__ rows, uint64_t runs) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < runs; i += (uint64_t)gridDim.x * blockDim.x) {
        uint4 w = rows[i];
        const MoveRow r = unpackMoveRow(w);
        uint64_t gap = MV_GAP_MAX;
        if (r.outRun < runs) {
            const uint64_t nx = rowIn(rows[r.outRun + 1]);
            gap = nx > r.out ? nx - r.out : 0;
            if (gap > MV_GAP_MAX) gap = MV_GAP_MAX;
        }
        w.w = (w.w & 0x07FFFFFFu) | ((uint32_t)gap << 27);
        rows[i] = w;
    }
}

// consistency of an unpacked table: what every walk relies on to stay inside it.  flags[0] counts violations.
__global__ void k_move_check(const uint4* __restrict__ rows, uint64_t runs, uint64_t n, uint32_t* __restrict__ flags) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= runs; i += (uint64_t)gridDim.x * blockDim.x) {
        const MoveRow r = unpackMoveRow(rows[i]);
        bool bad = false;
        if (i == runs) bad = r.in != n || r.head != 0; // the terminating row stops fast-forwards and forward walks
        else {
            const MoveRow nx = unpackMoveRow(rows[i + 1]);
            bad = r.head > 4 || (i == 0 && r.in != 0) || r.in >= nx.in || r.out >= n || r.outRun >= runs || r.out + (nx.in - r.in) > n;
            if (!bad) { // outputStartRun is the run that holds outputStartPos; the gap field says how far the next run is
                const uint64_t a = rowIn(rows[r.outRun]), b = rowIn(rows[r.outRun + 1]);
                bad = !(a <= r.out && r.out < b);
                if (!bad) bad = rowGap(rows[i]) != (uint32_t)(b - r.out > MV_GAP_MAX ? MV_GAP_MAX : b - r.out);
            }
        }
        if (bad) atomicAdd(&flags[0], 1u);
    }
}

// dir[b] = number of elements < (b << shift), b = 0 .. buckets (inclusive)
__global__ void k_posset_dir(const uint64_t* __restrict__ pos, uint64_t count, uint32_t shift, uint64_t buckets, uint64_t* __restrict__ dir) {
    for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b <= buckets; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t x = b << shift;
        uint64_t lo = 0, hi = count;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (pos[mid] < x) lo = mid + 1;
            else hi = mid;
        }
        dir[b] = lo;
    }
}
// a directory that arrived from elsewhere: the same values k_posset_dir would write
__global__ void k_posset_dir_check(const uint64_t* __restrict__ pos, uint64_t count, uint32_t shift, uint64_t buckets, const uint64_t* __restrict__ dir,
                                   uint32_t* __restrict__ flags) {
    for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b <= buckets; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t x = b << shift, d = dir[b];
        if (d > count || (d < count && pos[d] < x) || (d > 0 && pos[d - 1] >= x)) atomicAdd(&flags[0], 1u);
    }
}
// strictly increasing?
__global__ void k_posset_check(const uint64_t* __restrict__ pos, uint64_t count, uint64_t limit, uint32_t* __restrict__ flags) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x)
        if (pos[i] >= limit || (i + 1 < count && pos[i] >= pos[i + 1])) atomicAdd(&flags[0], 1u);
}
__global__ void k_run_map_check(const uint64_t* __restrict__ map, uint64_t count, uint64_t runs, uint32_t* __restrict__ flags) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x)
        if (map[i] >= runs) atomicAdd(&flags[0], 1u);
}

// a range as handed in by a caller: inside the table, run indices enclosing (valid ones exact)
__device__ inline bool rangeUsable(const MoveTable& t, uint64_t n, const MvRange& r) {
    if (!(r.begin < r.end && r.end <= n && r.beginRun <= r.endRun && r.endRun < t.runs)) return false;
    if (rowIn(t.rows[r.beginRun]) > r.begin || rowIn(t.rows[r.endRun + 1]) < r.end) return false; // enclosing
    if (r.valid && (rowIn(t.rows[r.beginRun + 1]) <= r.begin || rowIn(t.rows[r.endRun]) > r.end - 1)) return false; // exact
    return true;
}

// hook: all four children of every parent (cmb_move_extend_batch).  bad[0] counts parents that are not usable ranges.
__global__ void k_move_extend(const MoveDev ix, const int mode, const MoveRangeRec* __restrict__ parents, uint64_t n,
                              MoveRangeRec* __restrict__ children, uint8_t* __restrict__ ok, uint32_t* __restrict__ bad) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const MvPair parent = loadPair(parents[i]);
        const bool fw = mode == 0;
        if (!rangeUsable(fw ? ix.rev : ix.fwd, ix.n, fw ? parent.rev : parent.sa)) {
            atomicAdd(&bad[0], 1u);
            continue;
        }
        MvPair ch[4];
        const uint32_t mask = moveChildren(ix, mode, parent, ch);
        for (int c = 0; c < 4; c++) {
            children[4 * i + c] = storePair(ch[c]);
            ok[4 * i + c] = mask >> c & 1u;
        }
    }
}

// hook: the text positions of every range (BMove::collectTextPositions, bmove.cpp:500-541, in the reference's order:
// the toehold's position, its phi chain while PLCP >= depth, then the phi^-1 chain).  A range of width w writes w
// positions at out[offsets[i] - offsetBase]; bad[0] counts ranges whose chain does not have exactly that length.
__global__ void k_move_locate(const MoveDev ix, const MoveRangeRec* __restrict__ ranges, uint64_t n, const uint64_t* __restrict__ offsets,
                              const uint64_t offsetBase, uint64_t* __restrict__ out, uint32_t* __restrict__ bad, const bool skipEmpty) {
    const uint64_t stop = ix.fwd.samplesLast[ix.fwd.runs - 1]; // getInitialToehold() + 1 (bmove.h:139-142)
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const MoveRangeRec q = ranges[i];
        const uint64_t width = q.end - q.begin, depth = q.depth;
        if (skipEmpty && width == 0) continue;
        uint64_t* o = out + (offsets[i] - offsetBase);
        const uint64_t firstPos = q.toehold - (q.repEnd ? depth - 1 : 0); // bmove.cpp:553-556
        if (depth == 0 || width == 0 || firstPos >= ix.n) {
            atomicAdd(&bad[0], 1u);
            continue;
        }
        uint64_t cnt = 0, cur = firstPos;
        o[cnt++] = cur;
        while (cnt < width && plcpAt(ix, cur) >= depth) {
            cur = movePhi(ix, cur);
            if (cur >= ix.n) break;
            o[cnt++] = cur;
        }
        bool exact = cur < ix.n && plcpAt(ix, cur) < depth; // the phi chain ended by itself
        cur = firstPos;
        while (exact && cur != stop) {
            cur = movePhiInverse(ix, cur);
            if (cur >= ix.n) {
                exact = false;
                break;
            }
            if (plcpAt(ix, cur) < depth) break;
            if (cnt == width) {
                exact = false;
                break;
            }
            o[cnt++] = cur;
        }
        if (!exact || cnt != width) atomicAdd(&bad[0], 1u);
    }
}


// MoveLFReprBP::findLF on a row that is already in registers
__device__ inline void moveLFRow(const MoveTable& t, const uint4 rowWord, uint64_t& pos, uint64_t& run) {
    const MoveRow r = unpackMoveRow(rowWord);
    const uint64_t off = pos - r.in;
    const uint32_t gap = rowGap(rowWord);
    pos = r.out + off;
    run = r.outRun;
    if (off < gap) return;
    if (gap < MV_GAP_MAX) run++;
    while (rowIn(t.rows[run + 1]) <= pos) run++;
}

// One range with its toehold extended by one character: BMove::findRangeWithExtraCharBackward (bmove.cpp:299-326) over
// MoveLFReprBP::addChar (moverepr.cpp:309-327).  The range's run indices are exact on entry and on exit; the toehold never
// represents the end on this path.  Returns false (range untouched) if the character does not occur in the range.
__device__ inline bool moveExtendOne(const MoveTable& t, const uint32_t c, MvRange& r, uint64_t& toehold) {
    uint64_t run1 = r.beginRun, pos1 = r.begin;
    uint4 row1 = t.rows[run1];
    while (rowHead(row1) != c) { // walkToNextRun
        if (run1 == r.endRun) return false;
        run1++;
        row1 = t.rows[run1];
        pos1 = rowIn(row1);
    }
    uint64_t run2 = r.endRun, pos2 = r.end - 1;
    uint4 row2 = t.rows[run2];
    while (rowHead(row2) != c) { // walkToPreviousRun: ends at run1 at the latest
        pos2 = rowIn(row2) - 1;
        run2--;
        row2 = t.rows[run2];
    }
    const uint64_t parentWidth = r.end - r.begin, lastRun = run2, endRun = r.endRun;
    moveLFRow(t, row1, pos1, run1);
    moveLFRow(t, row2, pos2, run2);
    if (pos2 + 1 - pos1 == parentWidth) toehold -= 1; // toehold - !toeholdRepresentsEnd
    else toehold = (lastRun == endRun ? t.samplesFirst[endRun] : t.samplesLast[lastRun]) - 1; // computeToehold (bmove.cpp:222-243)
    r.begin = pos1, r.end = pos2 + 1, r.beginRun = run1, r.endRun = run2;
    return true;
}

// character codes of a read as Read / ReadBundle clean it (reads.h:43-58): upper case, anything outside ACGT is N (0 here)
__device__ inline uint32_t readCode(uint8_t ch) {
    ch &= 0xDF; // upper case
    return ch == 'A' ? 1u : ch == 'C' ? 2u : ch == 'G' ? 3u : ch == 'T' ? 4u : 0u;
}

// k = 0 on the b-move index: IndexInterface::exactMatchesOutput (indexinterface.cpp:947-1014, RLC branch) of every read
// (task 2 i) and of its reverse complement (task 2 i + 1): the range of the whole string by backward extension, with its
// toehold.  widths[task] = number of occurrences; nodes = NODE_COUNTER.
__global__ void k_move_exact(const MoveDev ix, const uint8_t* __restrict__ reads, const uint64_t* __restrict__ readOff, uint64_t nTasks,
                             MoveRangeRec* __restrict__ ranges, uint64_t* __restrict__ widths, unsigned long long* __restrict__ nodes) {
    unsigned long long myNodes = 0, myExp = 0; // NODE_COUNTER; extensions attempted (the byte model's unit: nodes[1])
    for (uint64_t task = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; task < nTasks; task += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t rd = task >> 1, b = readOff[rd], len = readOff[rd + 1] - b;
        const bool rc = task & 1;
        MvRange r = {0, ix.n, 0, ix.fwd.runs - 1, true}; // getCompleteRange (bmove.h:369-373)
        uint64_t toehold = ix.fwd.samplesLast[ix.fwd.runs - 1] - 1;
        bool alive = len > 0;
        for (uint64_t step = 0; alive && step < len; step++) {
            // the string is matched from its last character to its first; the reverse complement's character len-1-step
            // is the complement of the read's character `step` (nucleotide.h:250)
            uint32_t c = readCode(reads[b + (rc ? step : len - 1 - step)]);
            if (rc && c) c = 5 - c;
            myExp += c != 0;
            alive = c != 0 && moveExtendOne(ix.fwd, c, r, toehold);
            myNodes += alive;
        }
        MvPair p;
        p.sa = alive ? r : MvRange{0, 0, 0, 0, false};
        p.rev = MvRange{0, 0, 0, 0, true};
        p.toehold = alive ? toehold : 0, p.repEnd = false, p.depth = alive ? (uint32_t)len : 0;
        ranges[task] = storePair(p);
        widths[task] = alive ? r.end - r.begin : 0;
    }
    for (int o = 32; o; o >>= 1) {
        myNodes += __shfl_down(myNodes, o);
        myExp += __shfl_down(myExp, o);
    }
    if ((threadIdx.x & 63) == 0 && myExp) {
        atomicAdd(nodes, myNodes);
        atomicAdd(nodes + 1, myExp);
    }
}

struct MoveOccRec { // cmb_move_occ
    uint64_t begin, end;
    uint32_t distance, strand;
};
static_assert(sizeof(MoveOccRec) == 24, "record layout");

// positions of the located tasks -> occurrence records {begin, begin + read length, 0, strand}; one lane per occurrence
__global__ void k_move_occ(const uint64_t* __restrict__ positions, const uint64_t* __restrict__ taskOff, uint64_t nTasks, uint64_t total,
                           const uint64_t* __restrict__ readOff, MoveOccRec* __restrict__ out) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < total; j += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = 0, hi = nTasks; // the last task whose offset is <= j
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (taskOff[mid] <= j) lo = mid;
            else hi = mid;
        }
        const uint64_t rd = lo >> 1, len = readOff[rd + 1] - readOff[rd];
        out[j] = MoveOccRec{positions[j], positions[j] + len, 0u, (uint32_t)(lo & 1)};
    }
}
__global__ void k_move_read_offsets(const uint64_t* __restrict__ taskOff, uint64_t nReads, uint64_t* __restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= nReads; i += (uint64_t)gridDim.x * blockDim.x) out[i] = taskOff[2 * i];
}

// IndexInterface::populateTable of the RLC flavour (indexinterface.cpp:294-335): entry `key` = the ranges of the k-mer after
// wordSize forward extensions from the complete range, run indices of the SA range made exact (updateRangeSARuns); k-mers
// that do not occur keep SARangePair().  Key: 2 bits per character, the first character in the highest bits.
__global__ void k_move_kmer_table(const MoveDev ix, const uint32_t wordSize, MoveRangeRec* __restrict__ table) {
    const uint64_t total = 1ull << (2 * wordSize);
    for (uint64_t key = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; key < total; key += (uint64_t)gridDim.x * blockDim.x) {
        MvPair cur;
        cur.sa = {0, ix.n, 0, ix.fwd.runs - 1, true};
        cur.rev = {0, ix.n, 0, ix.rev.runs - 1, true};
        cur.toehold = ix.fwd.samplesLast[ix.fwd.runs - 1] - 1, cur.repEnd = false, cur.depth = 0;
        bool ok = true;
        for (int i = (int)wordSize - 1; i >= 0 && ok; i--) {
            const uint32_t c = (uint32_t)((key >> (2 * i)) & 3u);
            MvPair ch[4];
            const uint32_t mask = moveChildren(ix, 0, cur, ch);
            ok = mask >> c & 1u;
            cur = c == 0 ? ch[0] : c == 1 ? ch[1] : c == 2 ? ch[2] : ch[3];
        }
        if (ok) computeRunIndices(ix.fwd, cur.sa);
        else {
            cur.sa = {0, 0, 0, 0, true};
            cur.rev = cur.sa;
            cur.toehold = 0, cur.repEnd = false, cur.depth = 0;
        }
        table[key] = storePair(cur);
    }
}

} // namespace cmb

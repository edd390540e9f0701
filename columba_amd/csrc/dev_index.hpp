// Device-side view of the bidirectional FM-index and the rank / extend / locate primitives.
//
// Replaces (reference, paths relative to src/):
//   BitvecIntl<4>::rank              bitvec.h:356-372
//   BWTRepresentation<5>::occ/cumOcc fmindex/bwtrepr.h:80-107
//   FMIndex::findRangesWithExtraChar{Backward,Forward,BackwardUniDirectional}
//                                    fmindex/fmindex.cpp:137-243
//   FMIndex::findLF / findSA         fmindex/fmindex.cpp:47-60
//   Bitvec::rank (rank9)             bitvec.h:155-170
//
// HBM layout (round 1): the rank arrays keep the reference's interleaved layout, so the
// four cumulative bitvectors of one position share one 32-byte group and their eight
// L1/L2 count words share one 64-byte line; both are fetched with 16-byte vector loads
// (2 + 4 per position).  The BWT symbol needed by LF is decoded from the same 32-byte
// group (the bitvectors are cumulative: bwtrepr.h:67-68), so the 3-bit EncodedText
// (.bwt) is not kept on the device at all.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmb {

struct DevBWT {
    const uint64_t* bv;  // 4 words per 64 positions
    const uint64_t* cnt; // 8 words per 512 positions
    uint32_t dollarPos;
};

struct DevIndex {
    uint32_t n; // text length including '$'
    uint32_t counts[5];
    DevBWT fwd, rev;
    const uint64_t* saBv;
    const uint64_t* saCnt;
    const uint32_t* saSamples;
    const uint8_t* text;
    const uint4* kmer; // 4^kmerSize entries {sa.b, sa.e, rev.b, rev.e}
    uint32_t kmerSize;
    uint32_t switchPoint;
};

struct Range {
    uint32_t b, e;
    __host__ __device__ bool empty() const { return e <= b; }
    __host__ __device__ uint32_t width() const { return e <= b ? 0u : e - b; }
};
struct RangePair {
    Range sa, rev;
    __host__ __device__ bool empty() const { return sa.empty(); }
    __host__ __device__ uint32_t width() const { return sa.width(); }
};

// ranks of the four cumulative bitvectors at position p: R[i] = #{j < p : 1 <= BWT[j] <= i+1}
__device__ __forceinline__ void rank4(const DevBWT& t, uint32_t p, uint32_t R[4]) {
    const uint32_t w = p >> 6;
    const uint32_t b = p & 63u;
    const ulonglong2* cl = reinterpret_cast<const ulonglong2*>(t.cnt + (size_t)(p >> 9) * 8);
    const ulonglong2* bl = reinterpret_cast<const ulonglong2*>(t.bv + (size_t)w * 4);
    const ulonglong2 c0 = cl[0], c1 = cl[1], c2 = cl[2], c3 = cl[3];
    const ulonglong2 b0 = bl[0], b1 = bl[1];
    const uint32_t sub = w & 7u; // word within the 512-position block
    // bitvec.h:367-368: L2 partial of word `sub` (0 for the first word)
    const uint32_t sh = (sub == 0) ? 0u : (sub - 1u) * 9u;
    const uint64_t m = (sub == 0) ? 0ull : 0x1FFull;
    // bitvec.h:371: bits below position b
    const uint64_t lowmask = (b == 0) ? 0ull : (~0ull >> (64u - b));
    R[0] = (uint32_t)(c0.x + ((c0.y >> sh) & m)) + (uint32_t)__popcll(b0.x & lowmask);
    R[1] = (uint32_t)(c1.x + ((c1.y >> sh) & m)) + (uint32_t)__popcll(b0.y & lowmask);
    R[2] = (uint32_t)(c2.x + ((c2.y >> sh) & m)) + (uint32_t)__popcll(b1.x & lowmask);
    R[3] = (uint32_t)(c3.x + ((c3.y >> sh) & m)) + (uint32_t)__popcll(b1.y & lowmask);
}

// single rank(c, p) — test hook
__device__ __forceinline__ uint64_t rank1(const DevBWT& t, uint32_t c, uint64_t p) {
    uint64_t w = (p / 64) * 4 + c;
    uint64_t b = p % 64;
    uint64_t q = (p / 512) * 8 + 2 * c;
    uint64_t rv = t.cnt[q];
    uint64_t sub = (p / 64) % 8;
    if (sub) rv += (t.cnt[q + 1] >> ((sub - 1) * 9)) & 0x1FF;
    uint64_t lowmask = b ? (~0ull >> (64 - b)) : 0ull;
    return rv + __popcll(t.bv[w] & lowmask);
}

// occ(c,k), cumOcc(c,k) for c = 1..4 from the four cumulative ranks (bwtrepr.h:80-107)
__device__ __forceinline__ uint32_t occFromR(const uint32_t R[4], uint32_t c) {
    return c == 1 ? R[0] : R[c - 1] - R[c - 2];
}
__device__ __forceinline__ uint32_t cumFromR(const uint32_t R[4], uint32_t c, uint32_t dollarFlag) {
    return (c == 1 ? 0u : R[c - 2]) + dollarFlag;
}

// Extend `p` with character index c (1..4).  mode: 0 forward, 1 backward, 2 uni-directional
// backward.  Rb/Re are rank4 at the begin/end of the "trivial" range (p.sa for backward, p.rev
// for forward); db/de the (k > dollarPos) flags.
__device__ __forceinline__ bool childFromRanks(const DevIndex& ix, int mode, const RangePair& p,
                                               uint32_t c, const uint32_t Rb[4], const uint32_t Re[4],
                                               uint32_t db, uint32_t de, RangePair& child) {
    const uint32_t start = ix.counts[c];
    Range r1;
    r1.b = occFromR(Rb, c) + start;
    r1.e = occFromR(Re, c) + start;
    if (mode == 2) { // fmindex.cpp:226-243
        child.sa = r1;
        child.rev = Range{0u, 0u};
        return !r1.empty();
    }
    const uint32_t x = cumFromR(Re, c, de) - cumFromR(Rb, c, db);
    const uint32_t s = (mode == 1) ? p.rev.b : p.sa.b;
    Range r2;
    r2.b = s + x;
    r2.e = s + x + r1.width();
    if (mode == 1) { // backward fmindex.cpp:137-172
        child.sa = r1;
        child.rev = r2;
    } else { // forward fmindex.cpp:174-211
        child.sa = r2;
        child.rev = r1;
    }
    return !child.sa.empty();
}

// ranks needed to extend p in `mode`
__device__ __forceinline__ void loadExtendRanks(const DevIndex& ix, int mode, const RangePair& p,
                                                uint32_t Rb[4], uint32_t Re[4], uint32_t& db,
                                                uint32_t& de) {
    const DevBWT& t = (mode == 0) ? ix.rev : ix.fwd;
    const Range& tr = (mode == 0) ? p.rev : p.sa;
    rank4(t, tr.b, Rb);
    rank4(t, tr.e, Re);
    db = tr.b > t.dollarPos ? 1u : 0u;
    de = tr.e > t.dollarPos ? 1u : 0u;
}

// one-character extend (IndexInterface::addChar's inner call, indexinterface.cpp:1040)
__device__ __forceinline__ bool extendOne(const DevIndex& ix, int mode, const RangePair& p, uint32_t c,
                                          RangePair& child) {
    uint32_t Rb[4], Re[4], db, de;
    loadExtendRanks(ix, mode, p, Rb, Re, db, de);
    return childFromRanks(ix, mode, p, c, Rb, Re, db, de, child);
}

// rank9 Bitvec (bitvec.h:155-170)
__device__ __forceinline__ bool saMarked(const DevIndex& ix, uint32_t i) {
    return (ix.saBv[i >> 6] >> (i & 63u)) & 1ull;
}
__device__ __forceinline__ uint32_t saRank(const DevIndex& ix, uint32_t p) {
    const uint32_t w = p >> 6, b = p & 63u;
    const uint32_t q = (w >> 3) * 2;
    uint64_t rv = ix.saCnt[q];
    const uint32_t sub = w & 7u;
    if (sub) rv += (ix.saCnt[q + 1] >> ((sub - 1u) * 9u)) & 0x1FFull;
    const uint64_t lowmask = b ? (~0ull >> (64u - b)) : 0ull;
    return (uint32_t)rv + (uint32_t)__popcll(ix.saBv[w] & lowmask);
}

// findLF (fmindex.cpp:47-51): BWT symbol decoded from the cumulative bitvectors
__device__ __forceinline__ uint32_t findLF(const DevIndex& ix, uint32_t k) {
    const DevBWT& t = ix.fwd;
    if (k == t.dollarPos) {
        // symbol '$' (index 0): counts[0] + occ(0,k) = 0 + (k <= dollarPos ? 0 : 1) = 0
        return ix.counts[0];
    }
    uint32_t R[4];
    rank4(t, k, R);
    const ulonglong2* bl = reinterpret_cast<const ulonglong2*>(t.bv + (size_t)(k >> 6) * 4);
    const ulonglong2 b0 = bl[0], b1 = bl[1];
    const uint32_t bit = k & 63u;
    // smallest c with bit (c-1) set
    uint32_t c = ((b0.x >> bit) & 1ull) ? 1u : ((b0.y >> bit) & 1ull) ? 2u : ((b1.x >> bit) & 1ull) ? 3u : 4u;
    return ix.counts[c] + occFromR(R, c);
}

// findSA (fmindex.cpp:53-60); *lf accumulates the number of LF steps
__device__ __forceinline__ uint32_t findSA(const DevIndex& ix, uint32_t row, uint32_t* lf) {
    uint32_t l = 0;
    while (!saMarked(ix, row)) {
        row = findLF(ix, row);
        l++;
    }
    if (lf) *lf += l;
    return ix.saSamples[saRank(ix, row)] + l;
}

} // namespace cmb

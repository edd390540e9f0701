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
// HBM layout: the reference's two arrays per BWT (interleaved bitvector words + interleaved L1/L2
// counts, bitvec.h:209-232) are re-packed at index creation (k_relayout) into self-contained
// 32-byte rank blocks, one per 32 positions:
//     chunk 0      u32 abs[4]    cumulative rank of bitvector c at the block start (c = 0..2)
//     chunk 1      u32 bits[4]   the 32 bits of bitvectors 0..2 for the block's positions
// rank = abs + popcount(bits & lowmask): TWO 16-byte loads from one 32-byte sector (the reference layout
// needs a 64-byte count line plus a 32-byte bit group in another line).
// Bitvector 3 is not stored: every position but the '$' has its bit set (bwtrepr.h:57-68), so its rank at p is
// p - (p > dollarPos).  Its slots hold, in the blocks of the text's own BWT, the SPARSE SUFFIX ARRAY's bitvector instead:
// bits[3] = which of the block's 32 rows are sampled, abs[3] = the number of sampled rows before the block (the rank9
// structure of bitvec.h:155-170 folded in) — an LF step of findSA and its "is this row sampled, and which sample is it"
// come out of the SAME 32-byte sector (the reference reads the BWT word, two rank lines, the sampled-row word and its
// rank9 counts).  One byte per position and
// direction — 1.5 x the 128-byte blocks of 192 positions used before, which needed four loads per rank:
// a random fetch costs one 128-byte LINE of HBM traffic whatever it uses of it, and memory is not
// what an MI355X lacks.
// The BWT symbol needed by LF is decoded from the same bits (the bitvectors are cumulative:
// bwtrepr.h:67-68), so the 3-bit EncodedText (.bwt) is not kept on the device at all.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmb {

struct DevBWT {
    const uint4* blk; // 2 x 16 B per 32 positions (see above)
    uint32_t dollarPos;
};

struct DevIndex {
    uint32_t n; // text length including '$'
    uint32_t counts[5];
    DevBWT fwd, rev;
    const uint32_t* saSamples;
    const uint8_t* text;
    const uint32_t* text2; // 2 bits per character, 16 per word (nullptr if the text holds non-ACGT characters before '$')
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

constexpr uint32_t RANK_BLOCK = 32; // positions per rank block

struct RankChunks { // what one position needs from its block
    uint4 abs, bits;
    uint32_t bit;
};
__device__ __forceinline__ void loadRankChunks(const DevBWT& t, uint32_t p, RankChunks& k) {
    const uint4* B = t.blk + (size_t)(p >> 5) * 2;
    k.bit = p & 31u;
    k.abs = B[0];
    k.bits = B[1];
}
// (p, dollarPos: bitvector 3 is implicit — all positions but the '$')
__device__ __forceinline__ void ranksFromChunks(const RankChunks& k, uint32_t p, uint32_t dollarPos, uint32_t R[4]) {
    const uint32_t lowmask = (1u << k.bit) - 1u; // bitvec.h:371: bits below position `bit`
    R[0] = k.abs.x + (uint32_t)__popc(k.bits.x & lowmask);
    R[1] = k.abs.y + (uint32_t)__popc(k.bits.y & lowmask);
    R[2] = k.abs.z + (uint32_t)__popc(k.bits.z & lowmask);
    R[3] = p - (p > dollarPos ? 1u : 0u);
}

// the same as two raw 16-byte chunks {abs, bits} (callers that share the reply registers with other kinds of loads)
__device__ __forceinline__ void loadRankChunksRaw(const DevBWT& t, uint32_t p, uint4 v[2]) {
    const uint4* B = t.blk + (size_t)(p >> 5) * 2;
    v[0] = B[0];
    v[1] = B[1];
}
// both ends of a range: a narrow range often has both ends in ONE rank block — what the begin already fetched is not requested again.
// NOTHING here may read a reply: a copy `v[2] = v[0]` for the shared block made hipcc wait for the begin's chunks before it requested the
// end's (round 4: two dependent memory round trips per extension in k_parts, k_exact and the frontier kernels instead of one).  The
// consumer picks the end's chunks with rankPairEnd() once the replies are in.
__device__ __forceinline__ void loadRankPairRaw(const DevBWT& t, uint32_t pb, uint32_t pe, uint4 v[4]) {
    const uint32_t blkB = pb >> 5, blkE = pe >> 5;
    const uint4* B = t.blk + (size_t)blkB * 2;
    const uint4* E = t.blk + (size_t)blkE * 2;
    v[0] = B[0];
    v[1] = B[1];
    v[2] = v[3] = make_uint4(0u, 0u, 0u, 0u);
    if (blkE != blkB) {
        v[2] = E[0];
        v[3] = E[1];
    }
}
// the chunks {abs, bits} of the END of the range from what loadRankPairRaw requested
__device__ __forceinline__ void rankPairEnd(const uint4 v[4], uint32_t pb, uint32_t pe, uint4 w[2]) {
    const bool same = (pb >> 5) == (pe >> 5);
    w[0] = same ? v[0] : v[2];
    w[1] = same ? v[1] : v[3];
}
__device__ __forceinline__ void ranksFromRaw(const uint4 v[2], uint32_t p, uint32_t dollarPos, uint32_t R[4]) {
    const uint32_t lowmask = (1u << (p & 31u)) - 1u;
    R[0] = v[0].x + (uint32_t)__popc(v[1].x & lowmask);
    R[1] = v[0].y + (uint32_t)__popc(v[1].y & lowmask);
    R[2] = v[0].z + (uint32_t)__popc(v[1].z & lowmask);
    R[3] = p - (p > dollarPos ? 1u : 0u);
}

// ranks of the four cumulative bitvectors at position p: R[i] = #{j < p : 1 <= BWT[j] <= i+1}
// (BitvecIntl<4>::rank, bitvec.h:356-372, on the re-packed blocks)
__device__ __forceinline__ void rank4(const DevBWT& t, uint32_t p, uint32_t R[4]) {
    RankChunks k;
    loadRankChunks(t, p, k);
    ranksFromChunks(k, p, t.dollarPos, R);
}

// single rank(c, p) — test hook
__device__ __forceinline__ uint64_t rank1(const DevBWT& t, uint32_t c, uint64_t p) {
    uint32_t R[4];
    rank4(t, (uint32_t)p, R);
    return R[c];
}

// BitvecIntl<4>::rank on the reference's own layout (bitvec.h:356-372); used once, by k_relayout.
// p == N (one past the last position) is answered from position N-1 so that no word past the
// reference arrays is touched.
__device__ __forceinline__ uint32_t rankRefLayout(const uint64_t* bv, const uint64_t* cnt, uint32_t c, uint64_t p,
                                                  uint64_t N) {
    uint32_t extra = 0;
    if (p >= N) {
        p = N - 1;
        extra = (uint32_t)((bv[(p / 64) * 4 + c] >> (p % 64)) & 1ull);
    }
    const uint64_t q = (p / 512) * 8 + 2 * c;
    uint64_t rv = cnt[q];
    const uint64_t sub = (p / 64) % 8;
    if (sub) rv += (cnt[q + 1] >> ((sub - 1) * 9)) & 0x1FF;
    const uint64_t b = p % 64;
    const uint64_t lowmask = b ? (~0ull >> (64 - b)) : 0ull;
    return (uint32_t)rv + (uint32_t)__popcll(bv[(p / 64) * 4 + c] & lowmask) + extra;
}

// occ(c,k), cumOcc(c,k) for c = 1..4 from the four cumulative ranks (bwtrepr.h:80-107)
// (R[] is selected, not indexed: with a run-time c an indexed private array would live in scratch)
__device__ __forceinline__ uint32_t pickR(const uint32_t R[4], uint32_t i) {
    return i == 0 ? R[0] : i == 1 ? R[1] : i == 2 ? R[2] : R[3];
}
__device__ __forceinline__ uint32_t occFromR(const uint32_t R[4], uint32_t c) {
    return c == 1 ? R[0] : pickR(R, c - 1) - pickR(R, c - 2);
}
__device__ __forceinline__ uint32_t cumFromR(const uint32_t R[4], uint32_t c, uint32_t dollarFlag) {
    return (c == 1 ? 0u : pickR(R, c - 2)) + dollarFlag;
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

// The same without control flow (the frontier kernels are bound by instruction issue, and what hipcc makes of the three modes above is a
// nest of exec-mask saves and restores per child): both ranges are computed, the mode selects.
__device__ __forceinline__ bool childFromRanksFlat(const DevIndex& ix, int mode, const RangePair& p, uint32_t c, const uint32_t Rb[4],
                                                   const uint32_t Re[4], uint32_t db, uint32_t de, RangePair& child) {
    const uint32_t start = ix.counts[c];
    const uint32_t b1 = occFromR(Rb, c) + start, e1 = occFromR(Re, c) + start;
    const uint32_t w1 = e1 > b1 ? e1 - b1 : 0u;
    const uint32_t x = cumFromR(Re, c, de) - cumFromR(Rb, c, db);
    const uint32_t b2 = (mode == 1 ? p.rev.b : p.sa.b) + x, e2 = b2 + w1;
    const bool fwd = mode == 0, uni = mode == 2;
    child.sa.b = fwd ? b2 : b1;
    child.sa.e = fwd ? e2 : e1;
    child.rev.b = uni ? 0u : fwd ? b1 : b2;
    child.rev.e = uni ? 0u : fwd ? e1 : e2;
    return e1 > b1; // (the two ranges of a child have the same width)
}

// ranks needed to extend p in `mode`
__device__ __forceinline__ void loadExtendRanks(const DevIndex& ix, int mode, const RangePair& p,
                                                uint32_t Rb[4], uint32_t Re[4], uint32_t& db,
                                                uint32_t& de) {
    // values, not references, are selected: a select between two addresses keeps the objects in memory
    DevBWT t = ix.fwd;
    Range tr = p.sa;
    if (mode == 0) {
        t = ix.rev;
        tr = p.rev;
    }
    uint4 v[4], w[2];
    loadRankPairRaw(t, tr.b, tr.e, v);
    rankPairEnd(v, tr.b, tr.e, w);
    ranksFromRaw(v, tr.b, t.dollarPos, Rb);
    ranksFromRaw(w, tr.e, t.dollarPos, Re);
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

// The sparse suffix array (suffixArray.h:131-148, :229: a Bitvec over the rows with rank9, samples in row order) lives in
// slot 3 of the forward rank blocks (see the layout above).
__device__ __forceinline__ bool rowSampled(const RankChunks& k) { return (k.bits.w >> k.bit) & 1u; }
__device__ __forceinline__ uint32_t sampleIndex(const RankChunks& k) { // Bitvec::rank of the row (bitvec.h:155-170)
    return k.abs.w + (uint32_t)__popc(k.bits.w & ((1u << k.bit) - 1u));
}
__device__ __forceinline__ bool saMarked(const DevIndex& ix, uint32_t row) {
    RankChunks k;
    loadRankChunks(ix.fwd, row, k);
    return rowSampled(k);
}
__device__ __forceinline__ uint32_t saRank(const DevIndex& ix, uint32_t row) {
    RankChunks k;
    loadRankChunks(ix.fwd, row, k);
    return sampleIndex(k);
}

// findLF (fmindex.cpp:47-51) on a block that is already in registers: BWT symbol decoded from the cumulative bitvectors
__device__ __forceinline__ uint32_t lfFromChunks(const DevIndex& ix, const RankChunks& ch, uint32_t k) {
    if (k == ix.fwd.dollarPos) {
        // symbol '$' (index 0): counts[0] + occ(0,k) = 0 + (k <= dollarPos ? 0 : 1) = 0
        return ix.counts[0];
    }
    uint32_t R[4];
    ranksFromChunks(ch, k, ix.fwd.dollarPos, R);
    const uint32_t bit = ch.bit;
    // smallest c with bit (c-1) set
    const uint32_t c = ((ch.bits.x >> bit) & 1u) ? 1u : ((ch.bits.y >> bit) & 1u) ? 2u : ((ch.bits.z >> bit) & 1u) ? 3u : 4u;
    return ix.counts[c] + occFromR(R, c);
}
__device__ __forceinline__ uint32_t findLF(const DevIndex& ix, uint32_t k) {
    if (k == ix.fwd.dollarPos) return ix.counts[0];
    RankChunks ch;
    loadRankChunks(ix.fwd, k, ch);
    return lfFromChunks(ix, ch, k);
}

// findSA (fmindex.cpp:53-60); *lf accumulates the number of LF steps.  One 32-byte sector per visited row.
__device__ __forceinline__ uint32_t findSA(const DevIndex& ix, uint32_t row, uint32_t* lf) {
    uint32_t l = 0;
    RankChunks ch;
    loadRankChunks(ix.fwd, row, ch);
    while (!rowSampled(ch)) {
        row = lfFromChunks(ix, ch, row);
        loadRankChunks(ix.fwd, row, ch);
        l++;
    }
    if (lf) *lf += l;
    return ix.saSamples[sampleIndex(ch)] + l;
}

} // namespace cmb

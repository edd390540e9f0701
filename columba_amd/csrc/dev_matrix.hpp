// Device-side banded bit-parallel edit-distance matrix (64-bit words).
//
// Replaces BitParallelED<uint64_t> (reference src/bitparallelmatrix.h:300-750,
// src/bitparallelmatrix.cpp:34-123): setSequence / initializeMatrix / computeRow / at /
// inFinalColumn / onlyVerticalGapsLeft / getFirstColumn / findClusterCenters / traceBack.
//
// GPU formulation: the reference materialises, per part and direction, five match vectors
// per 32-row block (setSequence).  Here each read x strand carries two bit-strings per
// nucleotide (forward and reversed read); the 64-bit match word of ANY part, direction and
// block is a funnel-shifted window of those bit-strings (matchWord), so nothing is
// re-encoded per part and the same strings serve the full-read in-text matrix.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmb {

constexpr uint32_t MX_WORD = 64, MX_BLOCK = 32;
constexpr uint32_t MX_MAX_ED = 10; // bitparallelmatrix.h:313
constexpr uint32_t MX_LEFT = 21;   // :315
constexpr uint32_t MX_DIAG = 20;   // :316

// words per nucleotide bit-string: ceil(maxLen/32) + 3 zero words of padding
__host__ __device__ inline uint32_t gWords(uint32_t maxLen) { return (maxLen + 31) / 32 + 3; }

// bits [off, off+64) of bit-string G (off >= 0); words beyond nW read as 0 (caller pads)
__device__ __forceinline__ uint64_t window64(const uint32_t* G, uint32_t off) {
    const uint32_t w = off >> 5, sh = off & 31u;
    const uint64_t lo = (uint64_t)G[w] | ((uint64_t)G[w + 1] << 32);
    const uint64_t hi = G[w + 2];
    return sh ? ((lo >> sh) | (hi << (64u - sh))) : lo;
}

// Bit-string of nucleotide ch (0..3) of read x strand `rs`, of the sequence itself (reversed = 0) or of the reversed
// sequence (reversed = 1), in the per-READ layout k_prep writes (eight strings of gw words: [0..3] the read, [4..7]
// the reversed read).  The reverse-complement strand reads the other half with complemented nucleotide.
__device__ __forceinline__ const uint32_t* gString(const uint32_t* G, uint32_t gw, uint32_t rs, uint32_t reversed,
                                                   uint32_t ch) {
    const uint32_t strand = rs & 1u;
    const uint32_t s = 4u * ((reversed ^ strand) & 1u) + (strand ? 3u - ch : ch);
    return G + ((size_t)(rs >> 1) * 8 + s) * gw;
}

// Match word M for block b of a matrix whose horizontal sequence X (length xLen) starts at bit
// `xOff` of G (G = forward bit-string for FORWARD parts, reversed-read bit-string for BACKWARD
// parts).  Bit t of block b stands for column index j = t - LEFT + 32 b of X; the LEFT low bits of
// block 0 are forced to one (bitparallelmatrix.cpp:44-47); bits of columns >= xLen are zero.
template <uint32_t LEFT = MX_LEFT, uint32_t BLOCK = MX_BLOCK>
__device__ __forceinline__ uint64_t matchWord(const uint32_t* G, uint32_t xOff, uint32_t xLen, uint32_t b) {
    const int lim = (int)xLen + (int)LEFT - (int)(BLOCK * b); // number of meaningful low bits
    if (lim <= 0) return 0ull;
    // (with blocks of fewer rows than LEFT the forced ones reach into the blocks after the first: bitparallelmatrix.cpp:58 shifts them along)
    const int s = (int)LEFT - (int)(BLOCK * b);
    uint64_t m = s > 0 ? ((window64(G, xOff) << (uint32_t)s) | ((1ull << (uint32_t)s) - 1ull)) : window64(G, xOff + (uint32_t)(-s));
    if (lim < 64) m &= (1ull << lim) - 1ull;
    return m;
}

struct MatGeom {
    uint32_t n, m, Wv, Wh, maxED;
    __device__ __forceinline__ uint32_t sfc() const { return Wh + Wv + 1; } // :705
    __device__ __forceinline__ bool inFinalColumn(uint32_t i) const { return i >= m - sfc(); } // :437
    __device__ __forceinline__ uint32_t firstColumn(uint32_t i) const { return i <= Wv ? 0u : i - Wv; } // :670
};

// initializeMatrix (bitparallelmatrix.cpp:77-123).  The initial edit distances are
// initED[i] = raw[i] + increase for i in [0, nInit) (nInit >= 1 here; the reference's empty vector is
// only used by findCIGAR / the naive search); `first` / `last` are initED[0] / initED[nInit-1].  The
// first column is built from raw[] on the fly — no private array of initial distances is needed (it
// would live in scratch memory).
template <uint32_t LEFT = MX_LEFT, uint32_t DIAG = MX_DIAG, typename W = uint64_t>
__device__ __forceinline__ void initMatrix(MatGeom& g, uint32_t xLen, uint32_t maxED, uint32_t first, uint32_t last,
                                           const uint16_t* raw, uint32_t increase, uint32_t nInit, W& HP0,
                                           W& HN0, W& RAC0, uint32_t& score0) {
    g.n = xLen + 1;
    g.maxED = maxED;
    g.Wv = nInit - 1 + maxED - last;
    g.m = g.Wv + g.n;
    score0 = first;
    g.Wh = maxED - score0;
    if (g.Wv + g.Wh + 1 > g.m) g.m = g.Wv + g.Wh + 1;
    HP0 = (W)(~(W)0) << LEFT;
    HN0 = ~HP0;
    const uint32_t nn = nInit < LEFT + 1 ? nInit : LEFT + 1;
    for (uint32_t i = 1; i < nn; ++i) {
        const uint32_t cur = raw[i] + increase, prev = raw[i - 1] + increase; // length_t arithmetic
        if (cur < prev) {
            HP0 ^= (W)1 << (LEFT - i);
            HN0 ^= (W)1 << (LEFT - i);
        } else if (cur == prev) {
            HN0 ^= (W)1 << (LEFT - i);
        }
    }
    RAC0 = (W)1 << (DIAG + g.Wh);
}

// computeRow (bitparallelmatrix.h:352-415).  In/out: previous row state -> row i state.
// Returns false if every cell of row i exceeds maxED.
// The rightmost active column of the row just computed (bitparallelmatrix.h:400-412): called when D0 has no bit
// at the RAC column.  Returns false if every cell of the row exceeds maxED.
template <uint32_t BLOCK = MX_BLOCK, uint32_t DIAG = MX_DIAG>
__device__ __forceinline__ bool racWalk(const MatGeom& g, uint32_t i, uint64_t HP, uint64_t HN, uint64_t& RAC) {
    // The reference walks left from the RAC column, one column per iteration, until the running value
    // (1, -1 per HP bit, +1 per HN bit) reaches zero, and gives up at column diagBit - Wv (:400-412).  The walk
    // spans at most Wh + Wv <= 30 columns: it is done on 32-bit windows of HP / HN whose bit 31 is the RAC
    // column.  Almost always the first HP bit ends it (no HN bit before it): that case needs no loop.
    const uint32_t diagBit = i % BLOCK + DIAG;
    const uint32_t q = (uint32_t)__ffsll((unsigned long long)RAC) - 1u;
    const uint32_t maxSteps = q - (diagBit - g.Wv); // the walk fails if it is still running at this step
    uint32_t hp = (uint32_t)((HP << (63u - q)) >> 32); // bit 31 <- bit q (no branch on q)
    uint32_t hn = (uint32_t)((HN << (63u - q)) >> 32);
    const uint32_t p1 = hp ? (uint32_t)__clz(hp) : 32u; // steps before the first HP bit
    if (p1 >= maxSteps) return false;                    // (the value cannot reach zero before the stop column)
    uint32_t k = p1;
    if (p1 != 0u && (hn >> (32u - p1)) != 0u) {
        // HN bits before it: the general walk.  The value can only reach zero AT an HP bit, so the walk goes per HP bit,
        // not per column: the HN bits passed since the previous HP bit are counted, and it ends at the first HP bit at
        // which as many HP as HN bits (plus one) have been seen — before step maxSteps (zero reached AT the stop column
        // still fails, :408), or not at all.
        const uint32_t xs = maxSteps >= 32u ? hp : hp & ~(0xFFFFFFFFu >> maxSteps);
        uint32_t top = 0xFFFFFFFFu; // columns not yet passed
        int need = 1;               // HP bits still needed (rises with every HN bit passed)
        for (;;) {
            const uint32_t xm = xs & top;
            if (xm == 0u) return false;
            const uint32_t fh = (uint32_t)__builtin_clz(xm);
            const uint32_t hb = 0x80000000u >> fh;
            need += (int)__popc(hn & top & ~((hb << 1u) - 1u)) - 1;
            if (need == 0) {
                k = fh;
                break;
            }
            top = hb - 1u;
        }
    }
    RAC = 1ull << (q - k - 1u);
    return true;
}
// the RAC column moves with the window: one to the left per row, back by a block when the words are realigned
template <uint32_t BLOCK = MX_BLOCK>
__device__ __forceinline__ void racAdvance(uint32_t i, uint64_t& RAC) {
    RAC <<= 1u;
    if (i % BLOCK == 0) RAC >>= BLOCK;
}
__device__ __forceinline__ bool racHit(uint64_t D0, uint64_t RAC) { return (D0 & RAC) != 0ull; }
// the Hyyro recurrence of one row (bitparallelmatrix.h:352-398): no RAC, no score
template <uint32_t BLOCK = MX_BLOCK>
__device__ __forceinline__ void computeRowCore(uint32_t i, uint64_t M, uint64_t& HP, uint64_t& HN, uint64_t& D0) {
    if (i % BLOCK == 0) {
        HP >>= BLOCK;
        HN >>= BLOCK;
    }
    D0 = (((M & HP) + HP) ^ HP) | M | HN;
    const uint64_t VP = HN | ~(D0 | HP);
    const uint64_t VN = D0 & HP;
    HP = (VN << 1u) | ~(D0 | (VP << 1u));
    HN = (D0 & (VP << 1u));
}
template <uint32_t BLOCK = MX_BLOCK, uint32_t DIAG = MX_DIAG>
__device__ __forceinline__ bool computeRow(const MatGeom& g, uint32_t i, uint64_t M, uint64_t& HP, uint64_t& HN,
                                           uint64_t& D0, uint64_t& RAC, uint32_t& score) {
    racAdvance<BLOCK>(i, RAC);
    computeRowCore<BLOCK>(i, M, HP, HN, D0);
    score += (D0 & (1ull << (i % BLOCK + DIAG))) ? 0u : 1u;
    if (!racHit(D0, RAC)) return racWalk<BLOCK, DIAG>(g, i, HP, HN, RAC);
    return true;
}

// operator()(i,j) (bitparallelmatrix.h:622-639) from the state of row i
template <uint32_t BLOCK = MX_BLOCK, uint32_t DIAG = MX_DIAG>
__device__ __forceinline__ uint32_t cellAt(uint32_t i, uint32_t j, uint64_t HP, uint64_t HN, uint32_t score) {
    const uint32_t bit = (i % BLOCK) + DIAG;
    const uint32_t b = (i > j) ? bit - (i - j) + 1 : bit + 1;
    const uint32_t e = (i > j) ? bit + 1 : bit + (j - i) + 1;
    const uint32_t len = e - b;
    const uint64_t mask = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << b;
    const int neg = __popcll(HN & mask);
    const int pos = __popcll(HP & mask);
    return score + (uint32_t)((i > j) ? (neg - pos) : (pos - neg));
}

// ---- the same matrix on 32-bit words and 8-row blocks (in-text verification, maxED <= MX32_MAX_ED) ----
// computeRow / cellAt below are the reference's algorithm with WORD 32, BLOCK 8, DIAG 13, LEFT 14 instead of
// 64 / 32 / 20 / 21: half the VALU work per row.  Why the results (valid rows, RAC, every cell value <= maxED,
// and every "is this cell > maxED") are those of the 64-bit matrix: a cell of value <= maxED has |i - j| inside the
// band [-Wh, Wv] with Wv <= 3 maxED, and so has every cell of an optimal path to it; both windows contain the band
// in every row (here: bit 0 is at least 13 columns left of the diagonal, bit 31 at least 11 right of it).  What a
// window assumes about the cells outside (left: the cell keeps the value it had when its column left the window;
// right: new columns continue flat) only ever yields values above maxED — the left cell's value is at least
// 13 - (nZeros - 1) >= maxED + 1 exactly when maxED <= 4, the right one's at least 11 — and values only grow along
// paths, so in both windows: cells whose true value is <= maxED are exact, all others are > maxED.  Values above
// maxED may differ numerically between the windows; nothing reported depends on them (centres, distances and
// traceback only involve cells <= maxED and comparisons against them).  tests/test_gpu_parity.py compares counters
// (MATRIX_ROWS, ABORTED, CIGARS) and occurrences with the oracle's 64-bit matrix.
constexpr uint32_t MX32_BLOCK = 8, MX32_DIAG = 13, MX32_LEFT = 14, MX32_MAX_ED = 4;
// ---- ... and on 64-bit words and 16-row blocks (in-text verification, maxED 5 .. 7) ----
// BitParallelED<uint64_t> supports a first column of nZeros + maxED <= LEFT = 21 values: the in-text matrix of a
// candidate whose start is not fixed (nZeros = 2 k + 1) fits up to k = 6, and the reference switches to its 128-bit
// word at k = 7 (fmindex.h:240-246).  The same algorithm with WORD 64, BLOCK 16, DIAG 21, LEFT 22 holds the band of
// k = 7 (Wv = 3 k = 21 columns left of the diagonal, Wh = k right of it: bits 0 .. 44): same band, same values in the
// band, hence the same valid rows, centres and traces as the 128-bit matrix (pinned by the golden vectors of the
// reference's BitParallelED128 through the oracle; tests/test_gpu_parity.py compares at k = 5, 6, 7).
constexpr uint32_t MXW_BLOCK = 16, MXW_DIAG = 21, MXW_LEFT = 22, MXW_MAX_ED = 7;
// ---- ... and on 64-bit words and 16-row blocks with a wider left margin (in-text verification, maxED 8 .. 10) ----
// A candidate without a fixed start at k = 10 has Wv = 3 k = 30 columns left of the diagonal and Wh = k right of it: with BLOCK 16,
// DIAG 30, LEFT 31 the band occupies bits r % 16 .. r % 16 + 40 and the rightmost active column stays below bit 56 — the reference's
// own bound 3 maxED + 2 + BLOCK <= WORD (bitparallelmatrix.h:313) holds for 64 bits once the block is 16 rows.  Same band, same values in
// the band as the reference's 128-bit matrix (fmindex.h:240-246), by the argument above.  The walk to the rightmost active column can
// span Wv + Wh = 40 columns here: racWalkWide works on 64-bit windows.  Match words come straight from the read's bit-strings
// (matchWord<LEFT, BLOCK>: one funnel shift), not from the 32-row block words of k_match_words, whose 64 bits do not reach that far.
constexpr uint32_t MXX_BLOCK = 16, MXX_DIAG = 30, MXX_LEFT = 31;
// ... and for maxED 11 .. 13 (Wv = 39, Wh = 13): 8-row blocks, the band in bits r % 8 .. r % 8 + 52, the rightmost active column below bit 60
constexpr uint32_t MXY_BLOCK = 8, MXY_DIAG = 39, MXY_LEFT = 40;
// racWalk on 64-bit windows (bit 63 is the RAC column)
template <uint32_t BLOCK, uint32_t DIAG>
__device__ __forceinline__ bool racWalkWide(const MatGeom& g, uint32_t i, uint64_t HP, uint64_t HN, uint64_t& RAC) {
    const uint32_t diagBit = i % BLOCK + DIAG;
    const uint32_t q = (uint32_t)__ffsll((unsigned long long)RAC) - 1u;
    const uint32_t maxSteps = q - (diagBit - g.Wv); // the walk fails if it is still running at this step
    const uint64_t hp = HP << (63u - q), hn = HN << (63u - q);
    const uint32_t p1 = hp ? (uint32_t)__clzll((long long)hp) : 64u; // steps before the first HP bit
    if (p1 >= maxSteps) return false;
    uint32_t k = p1;
    if (p1 != 0u && (hn >> (64u - p1)) != 0ull) { // HN bits before it: per HP bit, as racWalk
        const uint64_t xs = maxSteps >= 64u ? hp : hp & ~(~0ull >> maxSteps);
        uint64_t top = ~0ull;
        int need = 1;
        for (;;) {
            const uint64_t xm = xs & top;
            if (xm == 0ull) return false;
            const uint32_t fh = (uint32_t)__clzll((long long)xm);
            const uint64_t hb = 0x8000000000000000ull >> fh;
            need += (int)__popcll(hn & top & ~((hb << 1u) - 1ull)) - 1;
            if (need == 0) {
                k = fh;
                break;
            }
            top = hb - 1ull;
        }
    }
    RAC = 1ull << (q - k - 1u);
    return true;
}
template <uint32_t BLOCK, uint32_t DIAG>
__device__ __forceinline__ bool computeRowWide(const MatGeom& g, uint32_t i, uint64_t M, uint64_t& HP, uint64_t& HN, uint64_t& D0, uint64_t& RAC,
                                               uint32_t& score) {
    racAdvance<BLOCK>(i, RAC);
    computeRowCore<BLOCK>(i, M, HP, HN, D0);
    score += (D0 & (1ull << (i % BLOCK + DIAG))) ? 0u : 1u;
    if (!racHit(D0, RAC)) return racWalkWide<BLOCK, DIAG>(g, i, HP, HN, RAC);
    return true;
}
// The match words of the full read (k_match_words: per 32-row block, bit t = character t - MXF_LEFT + 32 b) serve
// both in-text matrices: a row's word is a shift of its block's word.
// (23, not 22: with 22 the 32-bit matrix's shifts are whole bytes and the compiler turns the 64-bit LDS read of the
// block's word into an UNALIGNED 32-bit read — k_verify_edit 33.6 -> 43.5 ms)
constexpr uint32_t MXF_LEFT = 23;
__device__ __forceinline__ uint64_t matchWordW(uint64_t M64, uint32_t i) {
    return M64 >> (((i % MX_BLOCK) / MXW_BLOCK) * MXW_BLOCK + (MXF_LEFT - MXW_LEFT));
}
// the 32-bit match word of row i from the 64-bit word of the row's 32-row block (bit p' = bit p' + 8 s + 7)
__device__ __forceinline__ uint32_t matchWord32(uint64_t M64, uint32_t i) {
    return (uint32_t)(M64 >> (((i % MX_BLOCK) / MX32_BLOCK) * MX32_BLOCK + (MXF_LEFT - MX32_LEFT)));
}
// The rightmost active column of the 32-bit matrix: a one-bit mask, as in the 64-bit matrix.  The walk
// (bitparallelmatrix.h:400-412) goes left from the RAC column, one column per iteration, with a running value that
// starts at 1, falls at an HP bit and rises at an HN bit, until the value is zero; it fails if that has not happened
// above the stop column diagBit - Wv.  With x / y = the HP / HN bits from the RAC column leftwards (the two never share
// a bit): the walk fails iff x has no bit above the stop column (decided first: this is how most candidates end), ends
// at the highest bit of x when no HN bit lies above it (y <= x), and otherwise — the value can only reach zero AT an HP
// bit — is walked per HP bit, not per column: the HN bits passed since the previous HP bit are counted, and the walk
// ends at the first HP bit at which as many HP as HN bits (plus one) have been seen.
template <uint32_t BLOCK = MX32_BLOCK, uint32_t DIAG = MX32_DIAG>
__device__ __forceinline__ bool racWalk(const MatGeom& g, uint32_t i, uint32_t HP, uint32_t HN, uint32_t& rac) {
    const uint32_t l = i % BLOCK;
    const uint32_t below = (rac << 1u) - 1u; // the RAC column and everything left of it
    const uint32_t x = HP & below, y = HN & below;
    // highest bit of x above column l + DIAG - Wv  <=>  x >> (l + 1) >= 2^(DIAG - Wv)
    if ((x >> (l + 1u)) < (1u << (DIAG - g.Wv))) return false;
    if (y <= x) {
        rac = 0x40000000u >> (uint32_t)__builtin_clz(x); // (x != 0 here)
        return true;
    }
    const uint32_t xs = x & ~((2u << (l + DIAG - g.Wv)) - 1u); // zero reached AT the stop column still fails (:408)
    uint32_t top = below; // columns not yet passed
    int need = 1;         // HP bits still needed (rises with every HN bit passed)
    for (;;) {
        const uint32_t xm = xs & top;
        if (xm == 0u) return false;
        const uint32_t hb = 0x80000000u >> (uint32_t)__builtin_clz(xm);
        need += (int)__popc(y & top & ~((hb << 1u) - 1u)) - 1; // HN bits between the previous HP bit and this one
        if (need == 0) {
            rac = hb >> 1u;
            return true;
        }
        top = hb - 1u;
    }
}
template <uint32_t BLOCK = MX32_BLOCK>
__device__ __forceinline__ void racAdvance(uint32_t i, uint32_t& rac) {
    rac <<= 1u;
    if (i % BLOCK == 0) rac >>= BLOCK;
}
__device__ __forceinline__ bool racHit(uint32_t D0, uint32_t rac) { return (D0 & rac) != 0u; }
template <uint32_t BLOCK = MX32_BLOCK>
__device__ __forceinline__ void computeRowCore(uint32_t i, uint32_t M, uint32_t& HP, uint32_t& HN, uint32_t& D0) {
    if (i % BLOCK == 0) {
        HP >>= BLOCK;
        HN >>= BLOCK;
    }
    D0 = (((M & HP) + HP) ^ HP) | M | HN;
    const uint32_t VP = HN | ~(D0 | HP);
    const uint32_t VN = D0 & HP;
    HP = (VN << 1u) | ~(D0 | (VP << 1u));
    HN = (D0 & (VP << 1u));
}
template <uint32_t BLOCK = MX32_BLOCK, uint32_t DIAG = MX32_DIAG>
__device__ __forceinline__ bool computeRow(const MatGeom& g, uint32_t i, uint32_t M, uint32_t& HP, uint32_t& HN,
                                           uint32_t& D0, uint32_t& rac, uint32_t& score) {
    racAdvance<BLOCK>(i, rac);
    computeRowCore<BLOCK>(i, M, HP, HN, D0);
    score += (D0 >> (i % BLOCK + DIAG)) & 1u ? 0u : 1u;
    if (!racHit(D0, rac)) return racWalk<BLOCK, DIAG>(g, i, HP, HN, rac);
    return true;
}
// the RAC state of a matrix word type: a one-bit mask
__device__ __forceinline__ uint64_t racInit(uint64_t, uint32_t bit) { return 1ull << bit; }
__device__ __forceinline__ uint32_t racInit(uint32_t, uint32_t bit) { return 1u << bit; }
__device__ __forceinline__ uint32_t racIndex(uint64_t rac) { return (uint32_t)__ffsll((unsigned long long)rac) - 1u; }
__device__ __forceinline__ uint32_t racIndex(uint32_t rac) { return 31u - (uint32_t)__clz((int)rac); }
template <uint32_t BLOCK = MX32_BLOCK, uint32_t DIAG = MX32_DIAG>
__device__ __forceinline__ uint32_t cellAt(uint32_t i, uint32_t j, uint32_t HP, uint32_t HN, uint32_t score) {
    const uint32_t bit = (i % BLOCK) + DIAG;
    const uint32_t b = (i > j) ? bit - (i - j) + 1 : bit + 1;
    const uint32_t e = (i > j) ? bit + 1 : bit + (j - i) + 1;
    const uint32_t len = e - b;
    const uint32_t mask = (len >= 32 ? ~0u : ((1u << len) - 1u)) << b;
    const int neg = __popc(HN & mask);
    const int pos = __popc(HP & mask);
    return score + (uint32_t)((i > j) ? (neg - pos) : (pos - neg));
}

// onlyVerticalGapsLeft (bitparallelmatrix.h:651-665)
__device__ __forceinline__ bool onlyVerticalGapsLeft(const MatGeom& g, uint32_t i, uint64_t HN) {
    if (i + MX_LEFT < g.n) return false;
    const uint32_t b = i / MX_BLOCK;
    const uint32_t r = i % MX_BLOCK;
    const uint32_t bb = MX_DIAG - g.Wv + r + 1;
    const uint32_t be = MX_DIAG + g.n - b * MX_BLOCK;
    // Reference: (((~HN >> bb) << bb) << (WORD_SIZE - be)) == 0.  `be` can exceed 64 (up to 72)
    // when 44 < n - 32 b < 53; the reference then shifts by a negative count, which on the
    // x86-64 builds of Columba means "count mod 64".  Mirror that explicitly.
    const uint64_t v = (~HN >> bb) << bb;
    return (v << ((MX_WORD - be) & 63u)) == 0ull;
}


// ---- the in-index matrix at 11 ... 13 errors: 64-bit words, 16-row blocks ----
// The reference runs a search part on 64-bit words where the part's upper bound is at most 10 and on its 128-bit words beyond
// (indexinterface.cpp:391-398).  The frontier carries ONE matrix per node: BLOCK 16 gives the 64-bit word the reference's own bound of
// (64 - 16 - 2) / 3 = 15 errors (bitparallelmatrix.h:313), LEFT 31, DIAG 30 — the geometry of the in-text matrix above.  Cells of at
// most maxED are the same in every matrix that contains the band, hence valid rows, final-column values, cluster centres and first
// columns; `onlyVerticalGapsLeft` reads HN bits of cells that may exceed maxED and, in the reference, shifts by a negative count near
// the end of a block: it is answered as the part's OWN matrix would (refWord 64 or 128) — `true` where that shift count goes negative
// (be > refWord: the shifted word keeps only bits below bb, which are cleared), `false` where the last column lies beyond this word
// (it then lies beyond the band, and on a valid row — the only rows asked about, indexinterface.cpp:545 — a run of decreasing values
// cannot end there), the HN bits of columns i - Wv + 1 ... n - 1 otherwise.  oracle/: BitParallelEDT<uint64_t, 16> with `emulate`;
// tests/test_narrow_block_matrix.py and tools/soak_narrow_blocks.py (6.8e9 matrix rows) compare a search on it with the search on the
// reference's matrices — occurrences and every counter.
constexpr uint32_t MXN_BLOCK = 16, MXN_DIAG = 30, MXN_LEFT = 31, MXN_MAX_ED = 13;
template <uint32_t BLOCK, uint32_t DIAG>
__device__ __forceinline__ bool onlyVerticalGapsLeftAs(const MatGeom& g, uint32_t i, uint64_t HN, uint32_t refWord) {
    const uint32_t refBlock = refWord / 2u, refMaxED = (refWord - refBlock - 2u) / 3u, refLEFT = 2u * refMaxED + 1u, refDIAG = 2u * refMaxED;
    if (i + refLEFT < g.n) return false;
    if (refDIAG + g.n - (i / refBlock) * refBlock > refWord) return true;
    const uint32_t r = i % BLOCK;
    const int bb = (int)(DIAG - g.Wv + r + 1u);                 // column i - Wv + 1
    const int be = (int)(DIAG + r) + ((int)g.n - (int)i);       // one past column n - 1
    if (be > 64) return false;
    if (be <= bb) return true;
    const uint64_t mask = (be >= 64 ? ~0ull : ((1ull << be) - 1ull)) & ~((1ull << bb) - 1ull);
    return (~HN & mask) == 0ull;
}

// ---- the in-index matrix up to 6 errors: 32-bit words, 8-row blocks (round 4: GeoN32 of dev_bfs_edit.hpp) ----
// The frontier kernels are bound by instruction issue (profiles/r04_frontier_experiments.txt), and half of what they issue is the matrix
// row of a child on 64-bit words — register pairs, 64-bit shifts — of which the band of up to 7 errors uses a third.  The reference's own
// sizing rule (bitparallelmatrix.h:311-316) with WORD 32 and BLOCK 8 gives MATRIX_MAX_ED = (32 - 8 - 2) / 3 = 7, LEFT = 15, DIAG = 14: the
// band (Wv <= 14 columns left of the diagonal, Wh <= 7 right of it) lies in bits r % 8 .. r % 8 + 21, the rightmost active column below bit
// 29.  Cells of at most maxED are the same in every matrix that contains the band (see above), hence valid rows, final-column values,
// cluster centres and first columns; `onlyVerticalGapsLeft` is answered as the reference's 64-bit matrix would answer it
// (onlyVerticalGapsLeftAs, refWord 64).  As for the 16-row matrix of 11 ... 13 errors this was settled on the CPU before the device code
// was written: the oracle's search on BitParallelEDT<uint32_t, 8> against the search on the reference's matrices — occurrences and EVERY
// counter (tests/test_narrow_block_matrix.py, tools/soak_narrow32.py, tools/soak_narrow32_periodic.py).  The sizing rule is an upper
// bound at which the window has NO slack beside the band (Wv = 2 maxED = 14 = DIAG at 7 errors), and there the small matrix is not
// the reference's 64-bit one, which at 7 errors has six spare columns: on periodic texts (tandem repeats, the widest first columns and
// the most alternative alignments) a search at 7 errors computed a few matrix rows more or fewer in replays and once visited four
// nodes more — found by tools/soak_tiny_texts.py on the device, reproduced by the CPU experiment.  With two spare columns nothing
// differs (3 128 configurations on periodic texts, 300 on human-like ones): the small matrix serves batches of up to MXS_MAX_ED = 6
// errors, and a phase whose first column leaves fewer than two spare columns (Wv > DIAG - 2) raises FLAG_NARROW_MATRIX — the host
// then runs the batch on the 64-bit geometry.
// The match words stay those of the contexts (64-bit words of 32-row blocks, LEFT 21): bit p of the 8-row block a row lies in is bit
// p + 8 s + (21 - 15) of its 32-row block's word (s = the 8-row block's number within the 32 rows) — columns left of the sequence read as
// ones and columns beyond it as zeros in both.
constexpr uint32_t MXS_BLOCK = 8, MXS_DIAG = 14, MXS_LEFT = 15, MXS_MAX_ED = 6, MXS_SLACK = 2;
__device__ __forceinline__ uint32_t matchWordSmall(uint64_t M64, uint32_t i) {
    return (uint32_t)(M64 >> (((i % MX_BLOCK) / MXS_BLOCK) * MXS_BLOCK + (MX_LEFT - MXS_LEFT)));
}
template <uint32_t BLOCK, uint32_t DIAG>
__device__ __forceinline__ bool onlyVerticalGapsLeftAs(const MatGeom& g, uint32_t i, uint32_t HN, uint32_t refWord) {
    const uint32_t refBlock = refWord / 2u, refMaxED = (refWord - refBlock - 2u) / 3u, refLEFT = 2u * refMaxED + 1u, refDIAG = 2u * refMaxED;
    if (i + refLEFT < g.n) return false;
    if (refDIAG + g.n - (i / refBlock) * refBlock > refWord) return true;
    const uint32_t r = i % BLOCK;
    const int bb = (int)(DIAG - g.Wv + r + 1u);           // column i - Wv + 1
    const int be = (int)(DIAG + r) + ((int)g.n - (int)i); // one past column n - 1
    if (be > 32) return false;
    if (be <= bb) return true;
    const uint32_t mask = (be >= 32 ? ~0u : ((1u << be) - 1u)) & ~((1u << bb) - 1u);
    return (~HN & mask) == 0u;
}

// The two in-text matrices behind one set of names (k_verify_stage, forwardPass)
template <bool W32>
struct InTextMx;
template <>
struct InTextMx<true> {
    typedef uint32_t W;
    static constexpr uint32_t BLOCK = MX32_BLOCK, DIAG = MX32_DIAG, LEFT = MX32_LEFT;
    static __device__ __forceinline__ W matchWordOf(uint64_t M64, uint32_t i) { return matchWord32(M64, i); }
    static __device__ __forceinline__ void core(uint32_t i, W M, W& HP, W& HN, W& D0) { computeRowCore(i, M, HP, HN, D0); }
    static __device__ __forceinline__ void advance(uint32_t i, W& rac) { racAdvance(i, rac); }
    static __device__ __forceinline__ bool walk(const MatGeom& g, uint32_t i, W HP, W HN, W& rac) { return racWalk(g, i, HP, HN, rac); }
    static __device__ __forceinline__ bool row(const MatGeom& g, uint32_t i, W M, W& HP, W& HN, W& D0, W& rac, uint32_t& score) {
        return computeRow(g, i, M, HP, HN, D0, rac, score);
    }
    static __device__ __forceinline__ uint32_t cell(uint32_t i, uint32_t j, W HP, W HN, uint32_t score) { return cellAt(i, j, HP, HN, score); }
    static __device__ __forceinline__ uint32_t popc(W x) { return (uint32_t)__popc(x); }
};
template <>
struct InTextMx<false> {
    typedef uint64_t W;
    static constexpr uint32_t BLOCK = MXW_BLOCK, DIAG = MXW_DIAG, LEFT = MXW_LEFT;
    static __device__ __forceinline__ W matchWordOf(uint64_t M64, uint32_t i) { return matchWordW(M64, i); }
    static __device__ __forceinline__ void core(uint32_t i, W M, W& HP, W& HN, W& D0) { computeRowCore<BLOCK>(i, M, HP, HN, D0); }
    static __device__ __forceinline__ void advance(uint32_t i, W& rac) { racAdvance<BLOCK>(i, rac); }
    static __device__ __forceinline__ bool walk(const MatGeom& g, uint32_t i, W HP, W HN, W& rac) { return racWalk<BLOCK, DIAG>(g, i, HP, HN, rac); }
    static __device__ __forceinline__ bool row(const MatGeom& g, uint32_t i, W M, W& HP, W& HN, W& D0, W& rac, uint32_t& score) {
        return computeRow<BLOCK, DIAG>(g, i, M, HP, HN, D0, rac, score);
    }
    static __device__ __forceinline__ uint32_t cell(uint32_t i, uint32_t j, W HP, W HN, uint32_t score) { return cellAt<BLOCK, DIAG>(i, j, HP, HN, score); }
    static __device__ __forceinline__ uint32_t popc(W x) { return (uint32_t)__popcll(x); }
};

} // namespace cmb

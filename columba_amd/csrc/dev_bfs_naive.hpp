// Naive backtracking as a frontier (k_naive_start, k_naive_pass):
//   IndexInterface::approxMatchesNaive          reference src/indexinterface.cpp:1055-1141
//   IndexInterface::approxMatchesNaiveHamming   reference src/indexinterface.cpp:1143-1209
// What SearchStrategy::matchWithSearches runs instead of a search scheme for a read that is not longer than the
// number of parts (searchstrategy.cpp:148-152, :442-459) and for every read under `-S naive`
// (NaiveBackTrackingStrategy, searchstrategy.h:2785-2820: one part): the whole pattern is matched backward from the
// empty string with one banded matrix (first column 0, 1, 2, ...), EVERY node whose row lies in the final column
// with a value within the bound is an in-index occurrence (no cluster analysis), and a range that is not wider
// than the switch point goes to the in-text verification of the whole pattern at the range's OWN positions with a
// fixed start (:1120-1132; Hamming :1170-1175 with lengthBefore 0) — as the reference does it.
// The reference pops its stack depth first; the occurrence set, the work items and the counters do not depend on the
// order, so the frontier advances one row per pass like the other two searches.  A node is (ranges, read x strand,
// row, matrix row state | mismatches).  Reads that take this path are marked by k_parts (psel bit 7).
#pragma once
#include <type_traits>
// (included by kernels.hpp after dev_bfs_hamming.hpp)

namespace cmb {

struct NaiveBufs {
    uint4* Q[2];  // nodes, 3 planes of qCap: {ranges} {rsId, row | score << 16, RAC bit | mismatches << 8, -} {HP, HN}
    uint32_t qCap;
    uint32_t* nq; // [pass]
};

// the roots: one node per marked read x strand — the empty string's range at row 0 (getEmptyStringFMPos /
// getCompleteRange), the matrix of initializeMatrix(maxED) with an empty vector of initial distances
// (bitparallelmatrix.cpp:77-123: Wv = Wh = maxED, score 0)
__global__ void __launch_bounds__(256)
k_naive_start(DevIndex ix, const uint8_t* __restrict__ psel, const uint64_t* __restrict__ offs, uint32_t tasks,
              uint32_t k, uint32_t hamming, NaiveBufs B, Queues q, uint32_t narrow = 0 /* the matrix of 11 ... 13 errors (MXN_*) */) {
    const uint32_t rs = blockIdx.x * blockDim.x + threadIdx.x;
    bool root = rs < tasks && (psel[rs] & 0x80u) != 0u;
    if (root && hamming && offs[(rs >> 1) + 1] == offs[rs >> 1]) root = false; // (the reference indexes pattern[-1]: no defined result)
    uint32_t total;
    const uint32_t o = waveAppend(&B.nq[0], root ? 1u : 0u, total);
    if (!root) return;
    if (o >= B.qCap) {
        atomicOr(&q.cnt[3], (uint32_t)FLAG_NAIVE_Q);
        return;
    }
    const uint64_t HP0 = (~0ull) << (narrow ? MXN_LEFT : MX_LEFT);
    B.Q[0][o] = make_uint4(0u, ix.n, 0u, ix.n);
    B.Q[0][(size_t)B.qCap + o] = make_uint4(rs, 0u, (narrow ? MXN_DIAG : MX_DIAG) + k, 0u);
    B.Q[0][(size_t)2 * B.qCap + o] = make_uint4((uint32_t)HP0, (uint32_t)(HP0 >> 32), (uint32_t)~HP0, (uint32_t)(~HP0 >> 32));
}

// A block whose share of a queue does not fit drops it (the host grows the queue and runs the search again): the node slots
// it reserved stay unwritten, so the passes already launched behind it must not expand anything (one answer per block, as
// blockStopped in dev_bfs_edit.hpp).
constexpr uint32_t NAIVE_STOP = FLAG_NAIVE_Q | FLAG_ITEM_OVERFLOW | FLAG_FMOCC_OVERFLOW;
__device__ __forceinline__ bool naiveStopped(const Queues& q) {
    __shared__ uint32_t stopWord;
    if (threadIdx.x == 0) stopWord = __hip_atomic_load(&q.cnt[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & NAIVE_STOP;
    __syncthreads();
    return stopWord != 0u;
}

template <bool EDIT, bool NARROW = false /* the matrix of 11 ... 13 errors: 16-row blocks (dev_matrix.hpp: MXN_*) */>
__global__ void __launch_bounds__(256)
k_naive_pass(DevIndex ix, NaiveBufs B, uint32_t pass, const uint64_t* __restrict__ offs, uint32_t gw,
             const uint32_t* __restrict__ G, const uint8_t* __restrict__ seq, uint32_t maxLen, uint32_t k, Queues q) {
    using Mx = typename std::conditional<NARROW, MxNarrow, MxRef64>::type;
    __shared__ uint32_t sh[4][5];
    if (naiveStopped(q)) return;
    const uint32_t nIn = min(B.nq[pass], B.qCap);
    const uint4* __restrict__ Qi = B.Q[pass & 1u];
    uint4* __restrict__ Qo = B.Q[(pass + 1u) & 1u];
    const uint32_t qCap = B.qCap;
    const uint32_t sw = ix.switchPoint;
    uint32_t cNode = 0, cExp = 0, cRows = 0, flags = 0;
    for (uint32_t base = blockIdx.x * 256u; base < nIn; base += gridDim.x * 256u) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        uint4 cr[4];
        uint64_t cHP[4], cHN[4];
        uint32_t cState[4], cDist[4]; // row | score << 16 of the child, RAC bit | mismatches << 8; distance of its occurrence
        uint32_t kinds = 0;           // per child: bit 0 node, bit 1 in-text items, bit 2 in-index occurrence
        uint32_t rsId = 0, row1 = 0, nNode = 0, nIt = 0, nFm = 0;
        if (i < nIn) {
            const uint4 n0 = Qi[i], n1 = Qi[(size_t)qCap + i];
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
                const uint4 n2 = Qi[(size_t)2 * qCap + i];
                pHP = (uint64_t)n2.x | ((uint64_t)n2.y << 32);
                pHN = (uint64_t)n2.z | ((uint64_t)n2.w << 32);
            }
            const RangePair parent{{n0.x, n0.y}, {n0.z, n0.w}};
            uint32_t Rb[4], Re[4], db, de;
            loadExtendRanks(ix, 2, parent, Rb, Re, db, de); // extendFMPos, unidirectional backward (setDirection(BACKWARD, true))
            cExp++;
            row1 = row + 1;
            const uint32_t pc = EDIT ? 0u : seq[(size_t)rsId * maxLen + (len - row1)]; // pattern[size - row] (:1181)
#pragma unroll
            for (uint32_t ch = 1; ch <= 4; ch++) {
                RangePair child;
                if (!childFromRanks(ix, 2, parent, ch, Rb, Re, db, de, child)) continue;
                cNode++;
                cr[ch - 1] = make_uint4(child.sa.b, child.sa.e, child.rev.b, child.rev.e);
                const uint32_t width = child.sa.e - child.sa.b;
                uint32_t kd = 0;
                if (EDIT) {
                    if (row1 >= g.m) continue; // (:1098)
                    cRows++;
                    uint64_t HP = pHP, HN = pHN, D0, RAC = 1ull << rac;
                    uint32_t sc = score;
                    const uint64_t M = matchWord<Mx::LEFT, Mx::BLOCK>(gString(G, gw, rsId, 1u, ch - 1u), 0u, len, row1 / Mx::BLOCK);
                    if (!Mx::row(g, row1, M, HP, HN, D0, RAC, sc)) continue; // backtrack (:1104)
                    if (g.inFinalColumn(row1)) {
                        const uint32_t d = Mx::cell(row1, len, HP, HN, sc);
                        if (d <= k) {
                            kd |= 4u;
                            cDist[ch - 1] = d;
                        }
                    }
                    if (width <= sw) kd |= 2u; // crossing over to in-text verification (:1120)
                    else {
                        kd |= 1u;
                        cHP[ch - 1] = HP;
                        cHN[ch - 1] = HN;
                        cState[ch - 1] = row1 | (sc << 16);
                        cr[ch - 1].z = (uint32_t)__ffsll((unsigned long long)RAC) - 1u; // (the unused reverse range carries the RAC bit)
                    }
                } else {
                    if (width <= sw) kd = 2u; // checked first, when the node is popped (:1170)
                    else {
                        const uint32_t v1 = v + (ch != pc ? 1u : 0u);
                        if (v1 > k) continue;
                        if (row1 == len) {
                            kd = 4u;
                            cDist[ch - 1] = v1;
                        } else {
                            kd = 1u;
                            cState[ch - 1] = row1;
                            cr[ch - 1].z = v1 << 8;
                        }
                    }
                }
                kinds |= kd << (4 * (ch - 1));
                nNode += kd & 1u;
                nIt += (kd & 2u) ? width : 0u;
                nFm += (kd >> 2) & 1u;
            }
        }
        const uint32_t want[4] = {nNode, nIt, nFm, 0u};
        uint32_t got[4];
        blockAppend4(&B.nq[pass + 1], &q.cnt[0], &q.cnt[1], &q.cnt[1], want, sh, got);
        uint32_t oNode = got[0], oIt = got[1], oFm = got[2];
        bool ok = true;
        if (oNode + nNode > qCap) { ok = false; flags |= FLAG_NAIVE_Q; }
        if (oIt + nIt > q.itemCap) { ok = false; flags |= FLAG_ITEM_OVERFLOW; }
        if (oFm + nFm > q.fmCap) { ok = false; flags |= FLAG_FMOCC_OVERFLOW; }
        if (ok && kinds) {
            const uint32_t itMeta = EDIT ? packMeta(0u, k, 0u, 1u, ITEM_EDIT) : packMeta(0u, k, 0u, 0u, ITEM_HAMMING);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t kd = (kinds >> (4 * c)) & 15u;
                if (kd & 4u) q.fm[oFm++] = FMOccRec{rsId, cr[c].x, cr[c].y, row1, cDist[c], 0u};
                if (kd & 1u) {
                    const uint32_t racOrV = cr[c].z;
                    Qo[oNode] = make_uint4(cr[c].x, cr[c].y, 0u, 0u);
                    Qo[(size_t)qCap + oNode] = make_uint4(rsId, cState[c], racOrV, 0u);
                    if (EDIT)
                        Qo[(size_t)2 * qCap + oNode] = make_uint4((uint32_t)cHP[c], (uint32_t)(cHP[c] >> 32), (uint32_t)cHN[c],
                                                                  (uint32_t)(cHN[c] >> 32));
                    oNode++;
                } else if (kd & 2u) {
                    const uint32_t w = cr[c].y - cr[c].x;
                    for (uint32_t t = 0; t < w; t++) q.items[oIt + t] = make_uint4(rsId, cr[c].x + t, 0u, itMeta);
                    oIt += w;
                }
            }
        }
    }
    const uint32_t local[3] = {cNode, cExp, cRows};
    const int which[3] = {0, 7, 11}; // NODE_COUNTER, EXPANSIONS, MATRIX_ROWS
    flushCounters(q, local, which, 3);
    if (flags) atomicOr(&q.cnt[3], flags);
}

// ---- the fallback's own filter pass.  matchWithSearches hands the read to approxMatchesNaive[Hamming], which ends in
// getUniqueTextOccurrences / getTextOccHamming for THAT strand (indexinterface.cpp:1137, :1205); the survivors join the
// read's other occurrences as text occurrences (searchstrategy.cpp:455-457) and pass the filter of the mapping mode a
// second time.  So the raw text occurrences of the marked reads are filtered per read x strand first (the batch's own
// filter kernels with read x strand groups) and replaced by the survivors, which count as reported positions once more.
__global__ void __launch_bounds__(256)
k_naive_drop(TextOccRec* __restrict__ text, uint32_t n, const uint8_t* __restrict__ psel) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t rs = text[i].rsId;
    if (rs != 0xFFFFFFFFu && (psel[rs] & 0x80u)) text[i].rsId = 0xFFFFFFFFu;
}
__global__ void __launch_bounds__(256)
k_naive_keep(const unsigned long long* __restrict__ keys, uint32_t n, const uint64_t* __restrict__ offs, uint32_t k,
             const uint32_t* __restrict__ rank, const uint64_t* __restrict__ outOffs, TextOccRec* __restrict__ out, uint32_t layout) {
    const KeyBits kb = keyBits(layout);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t rk = rank[i];
    if (rk == 0xFFFFFFFFu) return;
    const unsigned long long key = keys[i]; // (k_pack_keys, read x strand groups)
    const uint32_t rs = (uint32_t)(key >> kb.group), r = rs >> 1;
    const uint32_t len = (uint32_t)(offs[r + 1] - offs[r]);
    const uint32_t begin = (uint32_t)(key >> kb.begin), dist = (uint32_t)(key >> kb.dist) & kb.distMask;
    const uint32_t width = layout == 1u ? len : len - k + ((uint32_t)(key >> 1) & kb.wMask);
    out[outOffs[rs] + rk] = TextOccRec{rs, begin, begin + width, dist};
}

} // namespace cmb

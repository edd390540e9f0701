// Hamming-distance search of one search scheme as a frontier (k_hbfs_*):
//   IndexInterface::recApproxMatchHamming          reference src/indexinterface.cpp:1211-1304
//   FMIndex::inTextVerificationHamming (dispatch)  reference src/fmindex/fmindex.cpp:409-428
// Same idea as dev_bfs_edit.hpp, without a matrix: a node is (ranges, row, mismatches so far) plus the few
// numbers of its phase that an expansion needs (part bounds, length before the part), so that the read
// character of the next row and the two rank blocks are fetched in ONE round trip.  A child that completes its
// part enters the next phase at once (the reference recurses there); the frontier advances one row per pass.
#pragma once
// (included by kernels.hpp after dev_bfs_edit.hpp: blockAppend4)
#include "dev_partition.hpp"

namespace cmb {

struct HbfsBufs {
    uint4* Q[2];   // nodes, 2 planes of qCap: {ranges} {rsId, scheme | search << 4 | idx << 9 | row << 13,
                   //                                   mismatches | startDepth << 8, pb | pe << 9 | lengthBefore << 18}
    uint32_t qCap;
    uint32_t* nq;  // [pass]
    unsigned long long* blockCnt; // [BFS_GRID][4]
};

// phase entry: what a node of phase `idx` carries (IndexInterface::recApproxMatchHamming prologue + the
// length before the pattern's leftmost processed part used by the in-text switch)
template <int MP>
__device__ __forceinline__ uint32_t hbfsPhaseWord(const DevSearchT<MP>& s, const PartOutT<MP>& po, int idx) {
    const int part = s.order[idx];
    const uint32_t lb = idx == 0 ? 0u : (uint32_t)po.pb[s.low[idx - 1]];
    return (uint32_t)po.pb[part] | ((uint32_t)po.pe[part] << 9) | (lb << 18);
}

template <bool START, int MP = MAXP>
__global__ void __launch_bounds__(256)
k_hbfs(DevIndex ix, const DevStrategyKT<MP>* __restrict__ stp, HbfsBufs B, uint32_t pass, const DfsTask* __restrict__ tasks,
       uint32_t nTasks, uint32_t maxLen, const uint8_t* __restrict__ seq, const PartOutT<MP>* __restrict__ parts, Queues q) {
    typedef DevStrategyKT<MP> DevStrategyK; // (the instance's table size)
    typedef DevSearchT<MP> DevSearch;
    typedef PartOutT<MP> PartOut;
    __shared__ uint32_t sh[4][5];
    extern __shared__ uint32_t stratLds[];
    if (blockStopped(q)) return;
    for (uint32_t i = threadIdx.x; i < sizeof(DevStrategyK) / 4; i += blockDim.x)
        stratLds[i] = reinterpret_cast<const uint32_t*>(stp)[i];
    __syncthreads();
    const DevStrategyK& st = *reinterpret_cast<const DevStrategyK*>(stratLds);
    const uint32_t outP = START ? 0u : pass + 1u;
    const uint32_t nIn = START ? nTasks : min(B.nq[pass], B.qCap);
    const uint4* __restrict__ Qi = B.Q[pass & 1u];
    uint4* __restrict__ Qo = B.Q[outP & 1u];
    const uint32_t qCap = B.qCap;
    const uint32_t sw = ix.switchPoint;
    uint32_t cNode = 0, cExp = 0, flags = 0;
    for (uint32_t base = blockIdx.x * 256u; base < nIn; base += gridDim.x * 256u) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        uint32_t kinds = 0;     // 4 bits per child: 1 node, 2 in-text items, 3 in-index occurrence
        uint4 cr[4];            // child ranges
        uint32_t cMeta[4], cVd[4], cPw[4]; // node words of the children that stay in the search
        uint32_t rsId = 0, itA = 0, itMeta = 0, fmDepth = 0;
        uint32_t nNode = 0, nIt = 0, nFm = 0;
        if (START) {
            if (i < nIn) {
                const DfsTask t = tasks[i];
                if (t.rsId != 0xFFFFFFFFu) { // (holes of the task queue)
                    const DevSearch& s = st.sch[t.scheme].s[t.search];
                    const PartOut po = parts[t.rsId];
                    rsId = t.rsId;
                    cr[0] = make_uint4(t.r.sa.b, t.r.sa.e, t.r.rev.b, t.r.rev.e);
                    cMeta[0] = (uint32_t)t.scheme | ((uint32_t)t.search << 4) | ((uint32_t)t.idx << 9);
                    cVd[0] = t.depth << 8;
                    cPw[0] = hbfsPhaseWord(s, po, t.idx);
                    kinds = 1;
                    nNode = 1;
                }
            }
        } else if (i < nIn) {
            const uint4 n0 = Qi[i], n1 = Qi[(size_t)qCap + i];
            rsId = n1.x;
            const uint32_t scheme = n1.y & 15u, search = (n1.y >> 4) & 31u, idx = (n1.y >> 9) & 15u, row = n1.y >> 13;
            const uint32_t v = n1.z & 0xFFu, smDepth = n1.z >> 8;
            const uint32_t pb = n1.w & 0x1FFu, pe = (n1.w >> 9) & 0x1FFu, lb = n1.w >> 18;
            const DevSearch& s = st.sch[scheme].s[search];
            const uint32_t dir = s.dir[idx];
            const int md = (s.uniAll || idx >= (uint32_t)s.uniIdx) ? 2 : (dir == 0 ? 0 : 1);
            const uint32_t xLen = pe - pb;
            // ---- the single memory step: the read character of the children's row + the rank blocks
            const uint32_t pc = seq[(size_t)rsId * maxLen + (dir == 0 ? pb + row : pe - row - 1)];
            const RangePair parent{{n0.x, n0.y}, {n0.z, n0.w}};
            uint32_t Rb[4], Re[4], db, de;
            loadExtendRanks(ix, md, parent, Rb, Re, db, de);
            cExp++;
            const uint32_t row1 = row + 1;
            itA = lb - (dir == 1 ? row1 : 0u);
            itMeta = packMeta(0, s.U[s.n - 1], s.L[s.n - 1], 0, ITEM_HAMMING);
            fmDepth = smDepth + xLen;
#pragma unroll
            for (uint32_t ch = 1; ch <= 4; ch++) {
                RangePair child;
                if (!childFromRanks(ix, md, parent, ch, Rb, Re, db, de, child)) continue;
                cNode++;
                cr[ch - 1] = make_uint4(child.sa.b, child.sa.e, child.rev.b, child.rev.e);
                if (child.sa.width() <= sw) { // in-text switch, checked when the node is popped (:1244)
                    kinds |= 2u << (4 * (ch - 1));
                    nIt += child.sa.width();
                    continue;
                }
                const uint32_t v1 = v + (ch != pc ? 1u : 0u);
                if (v1 > s.U[idx]) continue; // backtrack
                if (row1 == xLen) {          // end of the part
                    if (v1 < s.L[idx]) continue;
                    if (idx == (uint32_t)s.n - 1) {
                        kinds |= 3u << (4 * (ch - 1));
                        cVd[ch - 1] = v1;
                        nFm++;
                    } else { // recApproxMatchHamming(s, match, ..., idx + 1): the child is the start match
                        const PartOut po = parts[rsId];
                        kinds |= 1u << (4 * (ch - 1));
                        cMeta[ch - 1] = scheme | (search << 4) | ((idx + 1) << 9);
                        cVd[ch - 1] = v1 | (fmDepth << 8);
                        cPw[ch - 1] = hbfsPhaseWord(s, po, (int)idx + 1);
                        nNode++;
                    }
                    continue;
                }
                kinds |= 1u << (4 * (ch - 1));
                cMeta[ch - 1] = scheme | (search << 4) | (idx << 9) | (row1 << 13);
                cVd[ch - 1] = v1 | (smDepth << 8);
                cPw[ch - 1] = n1.w;
                nNode++;
            }
        }
        const uint32_t want[4] = {nNode, nIt, nFm, 0u};
        uint32_t got[4];
        blockAppend4(&B.nq[outP], &q.cnt[0], &q.cnt[1], &q.cnt[1], want, sh, got);
        uint32_t oNode = got[0], oIt = got[1], oFm = got[2];
        bool ok = true;
        if (oNode + nNode > qCap) { ok = false; flags |= FLAG_BFS_Q; }
        if (oIt + nIt > q.itemCap) { ok = false; flags |= FLAG_ITEM_OVERFLOW; }
        if (oFm + nFm > q.fmCap) { ok = false; flags |= FLAG_FMOCC_OVERFLOW; }
        if (ok) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t kd = (kinds >> (4 * c)) & 15u;
                if (kd == 1) {
                    Qo[oNode] = cr[c];
                    Qo[(size_t)qCap + oNode] = make_uint4(rsId, cMeta[c], cVd[c], cPw[c]);
                    oNode++;
                } else if (kd == 2) {
                    const uint32_t w = cr[c].y - cr[c].x;
                    for (uint32_t t = 0; t < w; t++) q.items[oIt + t] = make_uint4(rsId, cr[c].x + t, itA, itMeta);
                    oIt += w;
                } else if (kd == 3) {
                    q.fm[oFm++] = FMOccRec{rsId, cr[c].x, cr[c].y, fmDepth, cVd[c], 0u};
                }
            }
        }
    }
    unsigned long long v2[2] = {cNode, cExp};
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v2[j] += __shfl_xor(v2[j], d);
    }
    __shared__ unsigned long long shc[4][2];
    if ((threadIdx.x & 63u) == 0)
        for (int j = 0; j < 2; j++) shc[threadIdx.x >> 6][j] = v2[j];
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned long long t = shc[0][threadIdx.x] + shc[1][threadIdx.x] + shc[2][threadIdx.x] + shc[3][threadIdx.x];
        if (t) B.blockCnt[(size_t)blockIdx.x * 4 + threadIdx.x] += t;
    }
    if (flags) atomicOr(&q.cnt[3], flags);
}

} // namespace cmb

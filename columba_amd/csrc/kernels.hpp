// HIP kernels of the hot path (gfx950).  See DESIGN.md §Kernels for the per-kernel roofline.
//
//   k_kmer_table  populateTable                         src/indexinterface.cpp:294-335
//   k_rank/k_extend/k_locate   fine-grained hooks       bitvec.h:356, fmindex.cpp:137-243, :53-60
//   k_prep        read clean-up + reverse complement + match bit-strings
//                                                       src/reads.h:43-58, nucleotide.h:250,
//                                                       bitparallelmatrix.cpp:34-75
//   k_search      partition + search-scheme DFS         dev_search.hpp
//   k_verify      locate + in-text verification         fmindex.cpp:267-310, :358-407,
//                                                       indexhelpers.cpp:518-574, indexinterface.cpp:918-943
//   k_fmocc       in-index occurrence -> text positions indexinterface.cpp:1385-1440, :1349-1366
#pragma once
#include "dev_partition.hpp"

namespace cmb {

// ------------------------------------------------------------------ fine-grained hooks
__global__ void k_rank(DevIndex ix, int rev, const uint32_t* __restrict__ c, const uint64_t* __restrict__ p,
                       uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = rank1(rev ? ix.rev : ix.fwd, c[i], p[i]);
}

// all four children of each parent; one thread per parent.
__global__ void __launch_bounds__(256)
k_extend(DevIndex ix, int mode, const uint4* __restrict__ in, uint64_t n, uint4* __restrict__ out,
         uint8_t* __restrict__ ok) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        RangePair p{{v.x, v.y}, {v.z, v.w}};
        uint32_t Rb[4], Re[4], db, de;
        loadExtendRanks(ix, mode, p, Rb, Re, db, de);
        uint32_t okbits = 0;
#pragma unroll
        for (uint32_t ch = 1; ch <= 4; ch++) {
            RangePair child;
            const bool o = childFromRanks(ix, mode, p, ch, Rb, Re, db, de, child);
            out[i * 4 + (ch - 1)] = make_uint4(child.sa.b, child.sa.e, child.rev.b, child.rev.e);
            okbits |= (o ? 1u : 0u) << (8 * (ch - 1));
        }
        reinterpret_cast<uint32_t*>(ok)[i] = okbits;
    }
}

__global__ void k_locate(DevIndex ix, const uint32_t* __restrict__ rows, uint64_t n, uint32_t* __restrict__ out,
                         unsigned long long* lfTotal) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lf = 0;
    out[i] = findSA(ix, rows[i], &lf);
    if (lf) atomicAdd(lfTotal, (unsigned long long)lf);
}

// k-mer table: entry `key` = ranges of the k-mer after `kmerSize` forward extensions from the
// complete range; k-mers that do not occur keep SARangePair() = zeros (never inserted, :330).
__global__ void k_kmer_table(DevIndex ix, uint4* __restrict__ table) {
    const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = 1u << (2 * ix.kmerSize);
    if (key >= total) return;
    RangePair r{{0, ix.n}, {0, ix.n}};
    bool ok = true;
    for (int i = (int)ix.kmerSize - 1; i >= 0 && ok; i--) {
        const uint32_t code = ((key >> (2 * i)) & 3u) + 1u;
        RangePair child;
        ok = extendOne(ix, 0, r, code, child);
        r = child;
    }
    table[key] = ok ? make_uint4(r.sa.b, r.sa.e, r.rev.b, r.rev.e) : make_uint4(0, 0, 0, 0);
}

// ------------------------------------------------------------------ read preparation
// one thread per read x strand
__global__ void k_prep(const uint8_t* __restrict__ reads, const uint64_t* __restrict__ offs, uint32_t nReads,
                       uint32_t maxLen, uint32_t gw, uint8_t* __restrict__ seq, uint32_t* __restrict__ G) {
    const uint32_t rs = blockIdx.x * blockDim.x + threadIdx.x;
    if (rs >= 2 * nReads) return;
    const uint32_t r = rs >> 1;
    const bool rc = rs & 1u;
    const uint8_t* rd = reads + offs[r];
    const uint32_t len = (uint32_t)(offs[r + 1] - offs[r]);
    uint8_t* s = seq + (size_t)rs * maxLen;
    uint32_t* g = G + (size_t)rs * 8 * gw;
    for (uint32_t w = 0; w < 8 * gw; w++) g[w] = 0;
    if (len > maxLen) return;
    for (uint32_t i = 0; i < len; i++) {
        uint8_t ch = rc ? rd[len - 1 - i] : rd[i];
        ch &= 0xDF;
        uint32_t code = ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : ch == 'T' ? 4 : 5;
        if (rc && code <= 4) code = 5 - code;
        s[i] = (uint8_t)code;
        if (code <= 4) {
            g[(code - 1) * gw + (i >> 5)] |= 1u << (i & 31);
            const uint32_t ri = len - 1 - i;
            g[(4 + code - 1) * gw + (ri >> 5)] |= 1u << (ri & 31);
        }
    }
}

__device__ __forceinline__ void flushCounters(const Queues& q, const uint32_t* local, const int* which, int n) {
    for (int i = 0; i < n; i++)
        if (local[i]) atomicAdd(&q.counters[which[i]], (unsigned long long)local[i]);
}

// ------------------------------------------------------------------ prologue: the rank/extend kernel
// One lane per read x strand.  Every loop iteration performs at most ONE bidirectional extension per
// lane, at one common program point: 2 positions x (4 x 16 B of the 64-byte counts line + 2 x 16 B of
// the 32-byte bit group) = 12 independent 16-byte loads per lane in flight, whatever phase of the
// prologue the lane's read is in (dev_partition.hpp).
__global__ void __launch_bounds__(256, 4)
k_partition(DevIndex ix, const DevStrategyK* __restrict__ stp, const uint64_t* __restrict__ offs, uint32_t nReads,
            uint32_t k, uint32_t maxLen, const uint8_t* __restrict__ seq, PartOut* __restrict__ parts,
            DfsTask* __restrict__ dfsQ, uint32_t dfsCap, Queues q) {
    extern __shared__ uint32_t partLds[]; // 5 fields x numParts x blockDim.x words
    PartMachine m(ix, *stp, q, dfsQ, dfsCap, partLds, threadIdx.x, blockDim.x);
    const uint32_t total = 2 * nReads;
    uint32_t nextRs = blockIdx.x * blockDim.x + threadIdx.x;
    for (;;) {
        if (m.phase == PH_DONE) {
            uint32_t rs;
            if (q.dbg & 1u) {
                rs = nextRs;
                nextRs += gridDim.x * blockDim.x;
            } else {
                rs = atomicAdd(&q.cnt[4], 1u);
            }
            if (rs >= total) break;
            const uint32_t r = rs >> 1;
            m.begin(rs, (uint32_t)(offs[r + 1] - offs[r]), seq + (size_t)rs * maxLen, k);
        }
        m.advance();
        if (m.phase == PH_DONE && k > 0 && !(m.flags & FLAG_UNSUPPORTED_READ)) {
            PartOut po;
#pragma unroll
            for (int i = 0; i < MAXP; i++) {
                po.pb[i] = i < m.numParts ? (uint16_t)m.PB(i) : (uint16_t)0;
                po.pe[i] = i < m.numParts ? (uint16_t)m.PE(i) : (uint16_t)0;
            }
            parts[m.rsId] = po;
        }
        if (m.req) {
            RangePair child;
            const bool ok = extendOne(ix, m.reqMode, m.reqParent, m.reqCode, child);
            m.req = false;
            m.resume(ok, child);
        }
    }
    const uint32_t local[4] = {m.cNode, m.cExp, m.cImm, m.cStart};
    const int which[4] = {0, 7, 5, 6};
    flushCounters(q, local, which, 4);
    if (m.flags) atomicOr(&q.cnt[3], m.flags);
}

// ------------------------------------------------------------------ approximate DFS over the scheme
// One lane per DfsTask (a search whose exact start range is still wider than the switch point).
__global__ void __launch_bounds__(256)
k_dfs(DevIndex ix, const DevStrategyK* __restrict__ stp, const uint64_t* __restrict__ offs, uint32_t k,
      uint32_t maxLen, uint32_t gw, const uint8_t* __restrict__ seq, const uint32_t* __restrict__ G,
      const PartOut* __restrict__ parts, const DfsTask* __restrict__ tasks, uint32_t nTasks,
      Scratch* __restrict__ slabs, Queues q) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    Scratch& S = slabs[slot];
    Ctx c(ix, *stp, S, q);
    c.k = k;
    c.gw = gw;
    for (;;) {
        const uint32_t t = atomicAdd(&q.cnt[6], 1u);
        if (t >= nTasks) break;
        const DfsTask task = tasks[t];
        const uint32_t rs = task.rsId;
        c.rsId = rs;
        c.len = (uint32_t)(offs[(rs >> 1) + 1] - offs[rs >> 1]);
        c.seq = seq + (size_t)rs * maxLen;
        c.G = G + (size_t)rs * 8 * gw;
        const PartOut po = parts[rs];
#pragma unroll
        for (int i = 0; i < MAXP; i++) {
            S.pb[i] = po.pb[i];
            S.pe[i] = po.pe[i];
        }
        const DevSearch& s = stp->sch[task.scheme].s[task.search];
        if (stp->metric == 1) {
            OccTmp sm;
            sm.r = task.r;
            sm.dist = 0;
            sm.depth = task.depth;
            sm.shift = 0;
            sm.valid = true;
            EditSearch es(c, s);
            es.run(sm, task.idx);
        } else {
            HammingSearch hs(c, s);
            hs.run(task.r, task.depth, task.idx);
        }
    }
    const uint32_t local[3] = {c.cNode, c.cExp, c.cRows};
    const int which[3] = {0, 7, 11};
    flushCounters(q, local, which, 3);
    if (c.flags) atomicOr(&q.cnt[3], c.flags);
}

// ------------------------------------------------------------------ locate + verification
constexpr int VROWS = MAX_READ + 3 * 6 + 4;
struct VScratch {
    uint64_t HP[VROWS], HN[VROWS], D0[VROWS];
    uint16_t score[VROWS];
};

__device__ __forceinline__ void emitText(const Queues& q, uint32_t& flags, uint32_t rsId, uint32_t b, uint32_t e,
                                         uint32_t d) {
    const uint32_t base = atomicAdd(&q.cnt[2], 1u);
    if (base >= q.textCap) {
        flags |= FLAG_TEXT_OVERFLOW;
        return;
    }
    q.text[base] = TextOccRec{rsId, b, e, d};
}

__device__ __forceinline__ uint32_t textCode(uint8_t ch) { // A,C,G,T -> 0..3; anything else ('$') -> 4
    return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
}

__global__ void __launch_bounds__(256)
k_verify(DevIndex ix, const uint64_t* __restrict__ offs, uint32_t maxLen, uint32_t gw,
         const uint8_t* __restrict__ seq, const uint32_t* __restrict__ G, const uint4* __restrict__ items,
         uint32_t nItems, VScratch* __restrict__ vslabs, Queues q) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    VScratch& V = vslabs[slot];
    uint32_t cLF = 0, cLoc = 0, cText = 0, cRows = 0, cAbort = 0, cCig = 0, cStarted = 0, flags = 0;
    for (uint32_t it = slot; it < nItems; it += gridDim.x * blockDim.x) {
        const uint4 item = items[it];
        const uint32_t rs = item.x, row = item.y, a = item.z, meta = item.w;
        const uint32_t kind = (meta >> 21) & 3u;
        const uint32_t maxED = (meta >> 12) & 15u, minED = (meta >> 16) & 15u;
        const uint32_t fixed = (meta >> 20) & 1u;
        const uint32_t shift = meta & 0xFFFu;
        const uint32_t len = (uint32_t)(offs[(rs >> 1) + 1] - offs[rs >> 1]);
        const uint8_t* s = seq + (size_t)rs * maxLen;
        uint32_t pos;
        if ((meta >> 23) & 1u) { // direct start position (cmb_verify_batch hook): nothing to locate
            pos = row;
        } else {
            cLoc++;
            pos = findSA(ix, row, &cLF);
        }
        if (kind == ITEM_EXACT) { // verifyInTextExact (indexinterface.cpp:918-943)
            if (fixed) {
                emitText(q, flags, rs, pos, pos + len, 0);
                continue;
            }
            cStarted++;
            const uint32_t remaining = a;
            bool ok = pos >= remaining;
            const uint32_t p0 = pos - remaining;
            for (uint32_t j = 0; ok && j < remaining; j++) {
                cText++;
                if (textCode(ix.text[p0 + j]) + 1u != s[j]) ok = false;
            }
            if (ok) emitText(q, flags, rs, p0, p0 + len, 0);
            else cAbort++;
            continue;
        }
        if (kind == ITEM_HAMMING) { // FMIndex::inTextVerificationHamming (fmindex.cpp:370-406)
            cStarted++;
            const uint32_t lengthBefore = a;
            const uint32_t Tb = pos > lengthBefore ? pos - lengthBefore : 0;
            const uint32_t Te = Tb + len;
            if (Te > ix.n) continue;
            uint32_t score = 0;
            for (uint32_t j = 0; j < len; j++) {
                cText++;
                const uint32_t code = textCode(ix.text[Tb + j]) + 1u; // '$' -> 5 never equals a read code... 
                score += (code != s[j] || code > 4u);
                if (score > maxED) break;
            }
            if (score <= maxED && score >= minED) emitText(q, flags, rs, Tb, Te, score);
            continue;
        }
        // ---- edit distance: FMIndex::inTextVerification + InTextVerificationTask::doTask
        cStarted++;
        const uint32_t startDiff = a;
        const uint32_t sum = pos + shift; // getBeginPositions (fmindex.h:374-379)
        const uint32_t start = sum >= startDiff ? sum - startDiff : 0;
        const uint32_t nZeros = fixed ? 1u : 2u * maxED + 1u;
        MatGeom g;
        g.n = len + 1;
        g.maxED = maxED;
        g.Wv = nZeros - 1 + maxED;
        g.Wh = maxED;
        g.m = g.Wv + g.n;
        uint64_t HP = (~0ull) << MX_LEFT, HN = ~HP, D0 = 0, RAC = 1ull << (MX_DIAG + g.Wh);
        for (uint32_t i = 1; i < nZeros; i++) HN ^= 1ull << (MX_LEFT - i);
        uint32_t score = 0;
        V.HP[0] = HP;
        V.HN[0] = HN;
        V.score[0] = 0;
        const uint32_t maxEnd = ix.n - 1;
        const uint32_t hEnd = min(maxEnd, start + g.m - 1);
        const uint32_t size = hEnd > start ? hEnd - start : 0;
        if (!g.inFinalColumn(size)) continue;
        const uint32_t* Gf = G + (size_t)rs * 8 * gw;
        uint32_t i;
        uint64_t Mblk[4];
        for (i = 0; i < size; ++i) {
            const uint32_t r = i + 1;
            if ((r % MX_BLOCK) == 0 || i == 0) {
                const uint32_t b = r / MX_BLOCK;
#pragma unroll
                for (int ch = 0; ch < 4; ch++) Mblk[ch] = matchWord(Gf + ch * gw, 0, len, b);
            }
            const uint32_t tc = textCode(ix.text[start + i]);
            const uint64_t M = tc == 0 ? Mblk[0] : tc == 1 ? Mblk[1] : tc == 2 ? Mblk[2] : tc == 3 ? Mblk[3] : 0ull;
            cText++;
            cRows++;
            const bool valid = computeRow(g, r, M, HP, HN, D0, RAC, score);
            V.HP[r] = HP;
            V.HN[r] = HN;
            V.D0[r] = D0;
            V.score[r] = (uint16_t)score;
            if (!valid) break;
        }
        const uint32_t sfc = g.sfc();
        if (i <= size - sfc) { // u32 arithmetic as in the reference (indexhelpers.cpp:542)
            cAbort++;
            continue;
        }
        // findClusterCenters (bitparallelmatrix.h:591-614)
        const uint32_t lastRow = i;
        const uint32_t firstRow = (g.m - 1) - sfc;
        const uint32_t col = g.n - 1;
        uint32_t nCenters = 0;
        for (uint32_t ri = lastRow; ri > firstRow; ri--) {
            const uint32_t ED = cellAt(ri, col, V.HP[ri], V.HN[ri], V.score[ri]);
            if (ED > maxED || ED < minED) continue;
            const bool above = (ri == firstRow) || ED <= cellAt(ri - 1, col, V.HP[ri - 1], V.HN[ri - 1], V.score[ri - 1]);
            const bool below = (ri == lastRow) || ED <= cellAt(ri + 1, col, V.HP[ri + 1], V.HN[ri + 1], V.score[ri + 1]);
            if (!(above && below)) continue;
            nCenters++;
            // traceBack (bitparallelmatrix.h:531-586): only the begin offset is needed here
            uint32_t ti = ri, tj = col;
            while (tj > 0) {
                const uint32_t b = ti / MX_BLOCK;
                const uint64_t bit = 1ull << ((tj - b * MX_BLOCK) + MX_DIAG);
                if (V.HP[ti] & bit) {
                    --tj;
                } else {
                    bool diag = false;
                    if (ti > 0) {
                        const uint32_t tc = textCode(ix.text[start + ti - 1]);
                        const uint64_t M = tc < 4 ? matchWord(Gf + tc * gw, 0, len, b) : 0ull;
                        diag = ((M | ~V.D0[ti]) & bit) != 0;
                    }
                    if (diag) {
                        --ti;
                        --tj;
                    } else {
                        --ti;
                    }
                }
            }
            cCig++;
            emitText(q, flags, rs, start + ti, start + ri, ED);
        }
        if (nCenters == 0) cAbort++;
    }
    const uint32_t local[7] = {cLF, cLoc, cText, cRows, cAbort, cCig, cStarted};
    const int which[7] = {8, 9, 10, 11, 3, 4, 2};
    flushCounters(q, local, which, 7);
    if (flags) atomicOr(&q.cnt[3], flags);
}

// in-index occurrences (already de-duplicated per read) -> text occurrences
__global__ void __launch_bounds__(256)
k_fmocc(DevIndex ix, const FMOccRec* __restrict__ recs, uint32_t n, Queues q) {
    uint32_t cLF = 0, cLoc = 0, flags = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const FMOccRec f = recs[i];
        for (uint32_t row = f.b; row < f.e; row++) {
            cLoc++;
            const uint32_t p = findSA(ix, row, &cLF) + f.shift;
            emitText(q, flags, f.rsId, p, p + f.depth, f.dist);
        }
    }
    const uint32_t local[2] = {cLF, cLoc};
    const int which[2] = {8, 9};
    flushCounters(q, local, which, 2);
    if (flags) atomicOr(&q.cnt[3], flags);
}

} // namespace cmb

// HIP kernels of the hot path (gfx950).  See DESIGN.md §Kernels for the per-kernel roofline.
//
//   k_kmer_table  populateTable                         src/indexinterface.cpp:294-335
//   k_rank/k_extend/k_locate   fine-grained hooks       bitvec.h:356, fmindex.cpp:137-243, :53-60
//   k_prep        read clean-up + reverse complement + match bit-strings
//                                                       src/reads.h:43-58, nucleotide.h:250,
//                                                       bitparallelmatrix.cpp:34-75
//   k_parts / k_exact   partitioning, scheme selection, exact phases   dev_partition.hpp
//   k_bfs_start / k_bfs_pass / k_hbfs   the approximate search as a frontier   dev_bfs_edit.hpp, dev_bfs_hamming.hpp
//   k_verify      locate + in-text verification         fmindex.cpp:267-310, :358-407,
//                                                       indexhelpers.cpp:518-574, indexinterface.cpp:918-943
//   k_fmocc       in-index occurrence -> text positions indexinterface.cpp:1385-1440, :1349-1366
#pragma once
#include "dev_partition.hpp"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_run_length_encode.hpp>

namespace cmb {

// ------------------------------------------------------------------ fine-grained hooks
__global__ void k_rank(DevIndex ix, int rev, const uint32_t* __restrict__ c, const uint64_t* __restrict__ p,
                       uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = rank1(rev ? ix.rev : ix.fwd, c[i], p[i]);
}

// Bitvec::rank (rank9, bitvec.h:155-170) on the reference's own arrays; used once, by k_relayout.  Rows past the last
// word count everything.
__device__ __forceinline__ uint32_t rank9RefLayout(const uint64_t* bv, const uint64_t* cnt, uint64_t nWords, uint64_t p) {
    uint64_t w = p >> 6, b = p & 63;
    if (nWords == 0) return 0;
    if (w >= nWords) {
        w = nWords - 1;
        b = 64;
    }
    uint64_t rv = cnt[(w >> 3) * 2];
    const uint32_t sub = (uint32_t)(w & 7u);
    if (sub) rv += (cnt[(w >> 3) * 2 + 1] >> ((sub - 1u) * 9u)) & 0x1FFull;
    const uint64_t lowmask = b >= 64 ? ~0ull : b ? (~0ull >> (64 - b)) : 0ull;
    return (uint32_t)rv + (uint32_t)__popcll((unsigned long long)(bv[w] & lowmask));
}

// Re-pack one BWT's reference arrays into 32-byte rank blocks (dev_index.hpp); one thread per block.  Slot 3 (the
// bitvector that is all ones but for the '$') stays empty here; k_relayout_sa fills it in the forward table.
__global__ void k_relayout(const uint64_t* __restrict__ bv, const uint64_t* __restrict__ cnt, uint64_t N,
                           uint64_t nBlocks, uint4* __restrict__ out) {
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nBlocks) return;
    const uint64_t nWords = (N + 63) / 64;
    const uint64_t w = blk >> 1;
    const uint32_t sh = (uint32_t)(blk & 1u) * 32u;
    uint32_t bits[3];
    for (uint32_t c = 0; c < 3; c++) bits[c] = w < nWords ? (uint32_t)(bv[w * 4 + c] >> sh) : 0u;
    uint4 abs;
    abs.x = rankRefLayout(bv, cnt, 0, blk * RANK_BLOCK, N);
    abs.y = rankRefLayout(bv, cnt, 1, blk * RANK_BLOCK, N);
    abs.z = rankRefLayout(bv, cnt, 2, blk * RANK_BLOCK, N);
    abs.w = 0;
    out[blk * 2] = abs;
    out[blk * 2 + 1] = make_uint4(bits[0], bits[1], bits[2], 0u);
}

// The sparse suffix array's bitvector (suffixArray.h:131-148) and its rank9 counts into slot 3 of the forward rank blocks:
// bits.w = which of the block's 32 rows are sampled, abs.w = the number of sampled rows before the block.
__global__ void k_relayout_sa(const uint64_t* __restrict__ saBv, const uint64_t* __restrict__ saCnt, uint64_t saWords, uint64_t nBlocks,
                              uint4* __restrict__ out) {
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nBlocks) return;
    const uint64_t w = blk >> 1;
    const uint32_t sh = (uint32_t)(blk & 1u) * 32u;
    reinterpret_cast<uint32_t*>(out + blk * 2)[3] = rank9RefLayout(saBv, saCnt, saWords, blk * RANK_BLOCK);
    reinterpret_cast<uint32_t*>(out + blk * 2 + 1)[3] = w < saWords ? (uint32_t)(saBv[w] >> sh) : 0u;
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

// Consistency probe at index creation: findSA terminates because every LF walk reaches a sampled row within
// `sparseness` steps — IF the sampled-row bitvector, the samples and the BWT belong together.  With arrays that do not
// (a sparse suffix array of another sparseness or of another text) the walk of findSA would not end.  nProbe rows spread
// over the suffix array are walked with that bound; `bad` counts the rows that break it or leave the text.
__global__ void k_check_index(DevIndex ix, uint32_t maxSteps, uint32_t nProbe, uint32_t nSamples, uint32_t* __restrict__ bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nProbe) return;
    uint32_t row = (uint32_t)(((uint64_t)i * ix.n) / nProbe);
    for (uint32_t l = 0;; l++) {
        if (row >= ix.n) break;
        RankChunks ch;
        loadRankChunks(ix.fwd, row, ch);
        if (rowSampled(ch)) {
            const uint32_t si = sampleIndex(ch);
            if (si < nSamples && (uint64_t)ix.saSamples[si] + l < ix.n) return; // a position of the text: fine
            break;
        }
        if (l == maxSteps) break;
        row = lfFromChunks(ix, ch, row);
    }
    atomicAdd(bad, 1u);
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

// 16 text codes from ANY byte offset (gfx950 serves unaligned 16-byte global loads)
struct __attribute__((packed, aligned(1))) Unaligned16 {
    uint32_t x, y, z, w;
};
struct __attribute__((packed, aligned(1))) Unaligned8 {
    uint32_t x, y;
};
__device__ __forceinline__ uint4 loadText16(const uint8_t* p) {
    const Unaligned16 v = *reinterpret_cast<const Unaligned16*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// ------------------------------------------------------------------ read preparation
// One thread per (read, 32-character chunk): writes the codes of both strands and the chunk's word of
// the eight match bit-strings of both strands (forward / reversed read x A,C,G,T).  G must be zeroed
// beforehand (padding words stay zero).  Both strands need the same two byte runs of the read:
//   A = read[32w .. 32w+32)   and   B = read[L-1-32w-t], t = 0..31
__global__ void __launch_bounds__(256)
k_prep(const uint8_t* __restrict__ reads, const uint64_t* __restrict__ offs, uint32_t nReads, uint32_t maxLen,
       uint32_t gw, uint32_t chunks, uint8_t* __restrict__ seq, uint32_t* __restrict__ G, uint32_t* __restrict__ rec,
       uint32_t recW) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = gid / chunks, w = gid % chunks;
    if (r >= nReads) return;
    const uint8_t* rd = reads + offs[r];
    const uint32_t len = (uint32_t)(offs[r + 1] - offs[r]);
    if (len > maxLen) return;
    if (32 * w >= len) {
        if (w == 0 && rec) { // empty read: header only (rec is zeroed beforehand)
            rec[(size_t)(2 * r) * recW] = 0;
            rec[(size_t)(2 * r + 1) * recW] = 0;
        }
        return;
    }
    uint32_t fA[4] = {0, 0, 0, 0}, fB[4] = {0, 0, 0, 0}; // bit t set: A[t] / B[t] is that nucleotide
    uint8_t* sF = seq + (size_t)(2 * r) * maxLen;
    uint8_t* sR = seq + (size_t)(2 * r + 1) * maxLen;
    const uint32_t nT = min(32u, len - 32 * w);
    // the 32 codes of both strands are collected in registers and written as two 16-byte stores each (maxLen
    // is a multiple of 16): single-byte stores cost a partial-line write each
    uint32_t cF[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cR[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // the two byte runs, 32 bytes each, as four (unaligned) 16-byte loads instead of 64 byte loads: A forwards from
    // rd + 32 w (the buffer is padded behind its last read), B as the 32 bytes that END at rd + len - 32 w (B[t] is
    // byte 31 - t of them; they may begin in the previous read — only the first bytes of the buffer have nothing
    // in front of them: byte loads there)
    uint32_t wa[8], wb[8];
    {
        const uint4 x = loadText16(rd + 32 * w), y = loadText16(rd + 32 * w + 16);
        wa[0] = x.x, wa[1] = x.y, wa[2] = x.z, wa[3] = x.w, wa[4] = y.x, wa[5] = y.y, wa[6] = y.z, wa[7] = y.w;
    }
    if (offs[r] + len >= 32ull * w + 32ull) {
        const uint8_t* pb = rd + len - 32 * w - 32; // (pointer arithmetic inside the reads buffer)
        const uint4 x = loadText16(pb), y = loadText16(pb + 16);
        wb[0] = x.x, wb[1] = x.y, wb[2] = x.z, wb[3] = x.w, wb[4] = y.x, wb[5] = y.y, wb[6] = y.z, wb[7] = y.w;
    } else {
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) wb[j] = 0;
#pragma unroll
        for (uint32_t t = 0; t < 32; t++)
            if (t < nT) wb[(31u - t) >> 2] |= (uint32_t)rd[len - 1 - 32 * w - t] << (8u * ((31u - t) & 3u));
    }
#pragma unroll
    for (uint32_t t = 0; t < 32; t++) {
        if (t < nT) {
            // reads.h:43-58 (upper-case, non-ACGT -> N); nucleotide.h:250 (reverse complement keeps N)
            const uint8_t a = (uint8_t)(wa[t >> 2] >> (8u * (t & 3u))) & 0xDF;
            const uint8_t bch = (uint8_t)(wb[(31u - t) >> 2] >> (8u * ((31u - t) & 3u))) & 0xDF;
            const uint32_t ca = a == 'A' ? 1 : a == 'C' ? 2 : a == 'G' ? 3 : a == 'T' ? 4 : 5;
            const uint32_t cb = bch == 'A' ? 1 : bch == 'C' ? 2 : bch == 'G' ? 3 : bch == 'T' ? 4 : 5;
            cF[t >> 2] |= ca << (8 * (t & 3u));
            cR[t >> 2] |= (cb <= 4 ? 5 - cb : 5) << (8 * (t & 3u));
            if (ca <= 4) fA[ca - 1] |= 1u << t;
            if (cb <= 4) fB[cb - 1] |= 1u << t;
        }
    }
    {
        uint4* dF = reinterpret_cast<uint4*>(sF + 32 * w);
        uint4* dR = reinterpret_cast<uint4*>(sR + 32 * w);
        dF[0] = make_uint4(cF[0], cF[1], cF[2], cF[3]);
        dR[0] = make_uint4(cR[0], cR[1], cR[2], cR[3]);
        if (32 * w + 16 < maxLen) { // (the row ends at maxLen)
            dF[1] = make_uint4(cF[4], cF[5], cF[6], cF[7]);
            dR[1] = make_uint4(cR[4], cR[5], cR[6], cR[7]);
        }
    }
    // Eight bit-strings per READ: [0..3] the read's A,C,G,T bits, [4..7] those of the reversed read.  The
    // reverse-complement strand needs no strings of its own (gString): its position i is comp(read[L-1-i]), i.e.
    // string 4 + (3 - ch), and its reversed sequence is comp(read[ri]), i.e. string 3 - ch.
    uint32_t* gF = G + (size_t)r * 8 * gw;
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
        gF[ch * gw + w] = fA[ch];       // position i : read[i]
        gF[(4 + ch) * gw + w] = fB[ch]; // position ri: read[L-1-ri]
    }
    if (rec) {
        // read record for k_partition: word 0 = len | hasN << 16 (rec zeroed beforehand, chunks OR into it),
        // then per 32 characters the low / high bit of (code - 1): A=00 C=01 G=10 T=11
        uint32_t* rF = rec + (size_t)(2 * r) * recW;
        uint32_t* rR = rec + (size_t)(2 * r + 1) * recW;
        const uint32_t valid = nT >= 32 ? 0xFFFFFFFFu : ((1u << nT) - 1u);
        const uint32_t nA = valid & ~(fA[0] | fA[1] | fA[2] | fA[3]);
        const uint32_t nB = valid & ~(fB[0] | fB[1] | fB[2] | fB[3]);
        rF[1 + 2 * w] = fA[1] | fA[3];
        rF[2 + 2 * w] = fA[2] | fA[3];
        rR[1 + 2 * w] = fB[2] | fB[0]; // strand R position i is comp(read[L-1-i])
        rR[2 + 2 * w] = fB[1] | fB[0];
        const uint32_t hF = (w == 0 ? len : 0u) | (nA ? 0x10000u : 0u);
        const uint32_t hR = (w == 0 ? len : 0u) | (nB ? 0x10000u : 0u);
        if (hF) atomicOr(&rF[0], hF);
        if (hR) atomicOr(&rR[0], hR);
    }
}

} // namespace cmb
#include "dev_wave.hpp"
#include "dev_bfs_edit.hpp"
#include "dev_bfs_hamming.hpp"
#include "dev_bfs_naive.hpp"
namespace cmb {

// ---- the frontier kernels of the FM-index backend (the functions live in dev_bfs_edit.hpp, which the b-move translation
// unit includes as well)
__global__ void k_bfs_finish(BfsBufs B, Queues q) { // one block: per-block counters -> the batch counters
    __shared__ unsigned long long s[3];
    if (threadIdx.x < 3) s[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < BFS_GRID_CNT * 4; j += blockDim.x) {
        const unsigned long long v = B.blockCnt[j];
        if ((j & 3u) < 3u && v) atomicAdd(&s[j & 3u], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&q.counters[0], s[0]);  // NODE_COUNTER
        atomicAdd(&q.counters[7], s[1]);  // EXPANSIONS
        atomicAdd(&q.counters[12], s[1]); // DFS_EXPANSIONS
        atomicAdd(&q.counters[11], s[2]); // MATRIX_ROWS
    }
}


// first approximate phase of every task (k_exact's DfsTask queue) -> frontier of pass 0
template <class Geo = GeoN>
__global__ void __launch_bounds__(256)
k_bfs_start(DevIndex ix, const DevStrategyKT<Geo::MP>* __restrict__ stp, BfsBufs B, const DfsTask* __restrict__ tasks, uint32_t nTasks,
            const uint64_t* __restrict__ offs, uint32_t gw, const uint32_t* __restrict__ G, const PartOutT<Geo::MP>* __restrict__ parts,
            Queues q) {
    if (blockStopped(q)) return;
    bfsHeavy<true, FmTraits, Geo>(stp, B, 0u, tasks, nTasks, offs, gw, G, parts, q, blockIdx.x, gridDim.x);
}

// one level: blocks [0, BFS_GRID) expand the frontier, blocks [BFS_GRID, BFS_GRID + BFS_GRID_EV) handle the events
// of the same pass (both only append to the queues of pass + 1, so they run side by side)
#ifndef CMB_BFS_WAVES
#define CMB_BFS_WAVES 4 // wavefronts per SIMD k_bfs_pass is built for (128 VGPRs at most)
#endif
template <class Geo = GeoN>
__global__ void __launch_bounds__(256, Geo::MP == MAXP ? CMB_BFS_WAVES : 2)
k_bfs_pass(DevIndex ix, const DevStrategyKT<Geo::MP>* __restrict__ stp, BfsBufs B, uint32_t pass, const uint64_t* __restrict__ offs,
           uint32_t gw, const uint32_t* __restrict__ G, const PartOutT<Geo::MP>* __restrict__ parts, Queues q) {
    if (blockStopped(q)) return;
    if (blockIdx.x < B.gridX) bfsExpand<Geo>(ix, B, pass, q, blockIdx.x, B.gridX);
    else bfsHeavy<false, FmTraits, Geo>(stp, B, pass, nullptr, 0u, offs, gw, G, parts, q, blockIdx.x - B.gridX, B.gridEv);
}


// ------------------------------------------------------------------ prologue: the rank/extend kernels
// extension of `parent` by `code` from the raw chunks of its two rank blocks (loaded in the memory step)
__device__ __forceinline__ void issueExtend(const DevIndex& ix, int mode, const RangePair& parent, uint4 v[4]) {
    DevBWT t = ix.fwd;
    Range tr = parent.sa;
    if (mode == 0) {
        t = ix.rev;
        tr = parent.rev;
    }
    loadRankPairRaw(t, tr.b, tr.e, v);
}
__device__ __forceinline__ bool takeExtend(const DevIndex& ix, int mode, const RangePair& parent, uint32_t code,
                                           const uint4 v[4], RangePair& child) {
    DevBWT t = ix.fwd;
    Range tr = parent.sa;
    if (mode == 0) {
        t = ix.rev;
        tr = parent.rev;
    }
    uint32_t Rb[4], Re[4];
    uint4 w[2];
    rankPairEnd(v, tr.b, tr.e, w);
    ranksFromRaw(v, tr.b, t.dollarPos, Rb);
    ranksFromRaw(w, tr.e, t.dollarPos, Re);
    return childFromRanks(ix, mode, parent, code, Rb, Re, tr.b > t.dollarPos ? 1u : 0u, tr.e > t.dollarPos ? 1u : 0u,
                          child);
}

// Partitioning (dev_partition.hpp: PartMachine).  One lane per read x strand, static round-robin assignment
// (the partitioning of every read costs about the same).  Every loop iteration has ONE memory step: each lane
// issues the loads of its request — the two rank blocks of an extension (4 x 16 B from 2 sectors), the k-mer
// table entries of its seeds, or its read record — before any reply is consumed.
template <int PARTITION, bool LONG, int MP = MAXP>
__global__ void __launch_bounds__(256, MP == MAXP ? 4 : 2)
k_parts(DevIndex ix, const DevStrategyKT<MP>* __restrict__ stp, uint32_t nReads, uint32_t k, uint32_t maxLen,
        const uint8_t* __restrict__ seq, const uint4* __restrict__ rec, uint32_t recQ, PartOutT<MP>* __restrict__ parts,
        uint4* __restrict__ exr, uint8_t* __restrict__ psel, Queues q) {
    typedef DevStrategyKT<MP> DevStrategyK; // (the instance's table size)
    // LDS: a copy of the strategy tables, then per lane 5 x numParts partition words and 2 x ceil(maxLen/32)
    // read words
    extern __shared__ uint32_t partLds[];
    constexpr uint32_t STRAT_WORDS = (uint32_t)((sizeof(DevStrategyK) + 15) / 16 * 4);
    for (uint32_t i = threadIdx.x; i < sizeof(DevStrategyK) / 4; i += blockDim.x)
        partLds[i] = reinterpret_cast<const uint32_t*>(stp)[i];
    __syncthreads();
    const DevStrategyK& lst = *reinterpret_cast<const DevStrategyK*>(partLds);
    PartMachine<PARTITION, MP> m(ix, lst, partLds + STRAT_WORDS, threadIdx.x, blockDim.x);
    m.setReadWords(maxLen);
    const uint32_t total = 2 * nReads;
    // Lock-step batches: every lane of the wavefront takes one read x strand, all load their read records, all
    // fetch their seeds, then all run the extension loop (one extension per iteration, the memory step in the
    // middle) until the last one has assigned every character.  Reads of one batch need about the same number of
    // iterations, and only ONE phase's code runs at a time (lanes in different phases of a free-running state
    // machine made every iteration pay for all phases).
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t first = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t base = first & ~63u; base < total; base += stride) { // wave-uniform trip count
        const uint32_t rs = base + (threadIdx.x & 63u);
        uint4 v[MP];
        m.phase = PH_DONE;
        m.req = RQ_NONE;
        if (rs < total) {
#pragma unroll
            for (uint32_t j = 0; j < 5; j++)
                if (j < recQ) v[j] = rec[(size_t)rs * recQ + j];
            m.template begin<LONG>(rs, v, rec, recQ, seq + (size_t)rs * maxLen, k); // may leave a RQ_SEED request
            if (m.phase == PH_DONE) psel[rs] = 0x80u;      // unsupported read: nothing to search
        }
        if (__ballot(m.req == RQ_SEED) != 0ull) {
            if (m.req == RQ_SEED) {
                m.seedIssue(v);
                m.req = RQ_NONE;
                m.seedTake(v);
            }
        }
        for (;;) {
            if (m.req == RQ_NONE) m.advance(); // bookkeeping up to the next extension (or the end)
            if (__ballot(m.req == RQ_RANK) == 0ull) break;
            if (m.req == RQ_RANK) {
                uint4 rk[4]; // (its own array: the read-record path reinterprets v[] and would pin it in scratch)
                issueExtend(ix, m.reqMode, m.reqParent, rk);
                RangePair child;
                const bool ok = takeExtend(ix, m.reqMode, m.reqParent, m.reqCode, rk, child);
                m.req = RQ_NONE;
                m.resume(ok, child);
            }
        }
        if (m.phase == PH_FIN) m.finish(parts, exr, psel, total);
    }
    const uint32_t local[2] = {m.cNode, m.cExp};
    const int which[2] = {0, 7};
    flushCounters(q, local, which, 2);
    if (m.flags) atomicOr(&q.cnt[3], m.flags);
}

// Exact phases of the searches + part-level pre-verification (dev_partition.hpp: ExactLane); k = 0: the whole
// exact search.  One lane per (read x strand, slot), static round-robin; same one-memory-step loop.
template <bool LONG, int MP = MAXP>
__global__ void __launch_bounds__(256, MP == MAXP ? 4 : 2)
k_exact(DevIndex ix, const DevStrategyKT<MP>* __restrict__ stp, uint32_t nReads, uint32_t k, uint32_t maxLen,
        uint32_t nSlots, const uint8_t* __restrict__ seq, const uint4* __restrict__ rec, uint32_t recQ,
        const PartOutT<MP>* __restrict__ parts, const uint4* __restrict__ exr, const uint8_t* __restrict__ psel,
        DfsTask* __restrict__ dfsQ, uint32_t dfsCap, Queues q) {
    typedef DevStrategyKT<MP> DevStrategyK; // (the instance's table size)
    typedef DevSchemeT<MP> DevScheme;
    extern __shared__ uint32_t partLds[]; // strategy tables, then per lane numParts + 2 x ceil(maxLen/32) words
    constexpr uint32_t STRAT_WORDS = (uint32_t)((sizeof(DevStrategyK) + 15) / 16 * 4);
    for (uint32_t i = threadIdx.x; i < sizeof(DevStrategyK) / 4; i += blockDim.x)
        partLds[i] = reinterpret_cast<const uint32_t*>(stp)[i];
    __syncthreads();
    const DevStrategyK& lst = *reinterpret_cast<const DevStrategyK*>(partLds);
    ExactLane<MP> m(ix, lst, partLds + STRAT_WORDS, threadIdx.x, blockDim.x, maxLen);
    const uint32_t total = 2 * nReads;
    const uint64_t nTasks = (uint64_t)total * nSlots;
    uint64_t nextT = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int numParts = lst.numParts;
    const uint32_t sw = ix.switchPoint;
    bool done = false, ovI = false, ovD = false;
    WaveChunk chI, chD;
    auto holeI = [&](uint32_t i) { q.items[i] = make_uint4(0xFFFFFFFFu, 0, 0, 0); };
    auto holeD = [&](uint32_t i) { dfsQ[i].rsId = 0xFFFFFFFFu; };
    for (;;) {
        // (1) an idle lane takes its next (read x strand, slot)
        if (!done && m.phase == EX_IDLE && !m.req) {
            if (nextT >= nTasks) done = true;
            else {
                m.rsId = (uint32_t)(nextT / nSlots);
                m.slot = (uint32_t)(nextT % nSlots);
                m.seq = seq + (size_t)m.rsId * maxLen;
                m.phase = k == 0 ? EX_LOAD : EX_HDR;
                nextT += (uint64_t)gridDim.x * blockDim.x;
            }
        }
        // (2) bookkeeping of running searches up to their next extension
        if (!done && !m.req) {
            if (m.phase == EX_RUN) m.advance();
            else if (m.phase == EX_K0) m.advanceK0();
        }
        // (3) the memory step
        uint4 v[MP];
        uint32_t hdr = 0;
        const int ph = done ? EX_IDLE : m.phase;
        const bool isPost = m.slot == nSlots - 1;
        if (m.req) {
            issueExtend(ix, m.reqMode, m.cur, v);
        } else if (ph == EX_HDR) { // scheme selection + parts
            hdr = psel[m.rsId];
            const uint4* pp = reinterpret_cast<const uint4*>(parts + m.rsId);
#pragma unroll
            for (int j = 0; j < MP / 4; j++) v[j] = pp[j]; // (pb[MP], pe[MP]: 4 MP bytes)
        } else if (ph == EX_LOAD) {
            if (k == 0 || !isPost) { // read record (+ exact range of the first part of the search)
#pragma unroll
                for (uint32_t j = 0; j < 5; j++)
                    if (j < recQ) v[j] = rec[(size_t)m.rsId * recQ + j];
                if (k != 0) v[5] = exr[(size_t)lst.sch[m.sel].s[m.slot].order[0] * total + m.rsId];
            } else { // exact ranges of all parts
#pragma unroll
                for (int i = 0; i < MP; i++)
                    if (i < numParts) v[i] = exr[(size_t)i * total + m.rsId];
            }
        }
        if (m.req) {
            RangePair child;
            const bool ok = takeExtend(ix, m.reqMode, m.cur, m.reqCode, v, child);
            m.req = false;
            m.resume(ok, child);
        } else if (ph == EX_HDR) {
            m.sel = (int)(hdr & 0x7Fu);
            const uint16_t* pv = reinterpret_cast<const uint16_t*>(v);
#pragma unroll
            for (int i = 0; i < MP; i++)
                if (i < numParts) m.PBE(i) = (uint32_t)pv[i] | ((uint32_t)pv[MP + i] << 16);
            m.phase = EX_LOAD;
            if (hdr & 0x80u) m.phase = EX_IDLE; // unsupported read (reported by k_parts)
            else if (!isPost) {
                const DevScheme& sch = lst.sch[m.sel];
                if (m.slot >= sch.nSearches) m.phase = EX_IDLE;
                else if (sch.s[m.slot].U[0] > 0) { // recApproxMatchEditEntry on the complete range
                    if (lst.metric == 1) m.cStart++;
                    m.emitDfs(0, RangePair{{0, ix.n}, {0, ix.n}}, 0);
                    m.phase = EX_IDLE;
                }
            }
        } else if (ph == EX_LOAD) {
            if (k == 0) {
                m.template takeRecord<LONG>(v, rec, recQ);
                m.cur = RangePair{{0, ix.n}, {0, 0}};
                m.k0i = m.len;
                m.phase = m.len == 0 ? EX_IDLE : EX_K0;
            } else if (!isPost) {
                m.template takeRecord<LONG>(v, rec, recQ);
                m.startSearch(RangePair{{v[5].x, v[5].y}, {v[5].z, v[5].w}});
            } else {
                m.phase = EX_IDLE;
            }
        }
        // (4) part-level pre-verification (searchstrategy.cpp:464-476): one item group per narrow part
        if (__ballot(ph == EX_LOAD && k != 0 && isPost) != 0ull) {
            const bool mine = ph == EX_LOAD && k != 0 && isPost;
#pragma unroll
            for (int i = 0; i < MP; i++) {
                if (i >= numParts) break;
                uint32_t n = 0, a = 0, meta = 0;
                if (mine) {
                    const uint32_t width = v[i].y > v[i].x ? v[i].y - v[i].x : 0u;
                    if (width != 0 && width <= sw) {
                        n = width;
                        const uint32_t bg = m.PB(i);
                        if (lst.metric == 1) {
                            m.cImm++; // verifyExactPartialMatchInText (fmindex.cpp:253)
                            a = bg == 0 ? 0 : bg + k;
                            meta = packMeta(0, k, 0, bg == 0, ITEM_EDIT);
                        } else {
                            a = bg;
                            meta = packMeta(0, k, 0, 0, ITEM_HAMMING);
                        }
                    }
                }
                if (__ballot(n > 0) == 0ull) continue;
                const uint32_t o = chI.alloc(&q.cnt[0], q.itemCap, n, 256u, ovI, holeI);
                if (n && o != 0xFFFFFFFFu)
                    for (uint32_t t = 0; t < n; t++) q.items[o + t] = make_uint4(m.rsId, v[i].x + t, a, meta);
            }
        }
        // (5) queue what the searches staged
        if (__ballot(m.stN > 0) != 0ull) {
            const uint32_t o = chI.alloc(&q.cnt[0], q.itemCap, m.stN, 256u, ovI, holeI);
            if (m.stN && o != 0xFFFFFFFFu)
                for (uint32_t t = 0; t < m.stN; t++) q.items[o + t] = make_uint4(m.rsId, m.stB + t, m.stA, m.stMeta);
            m.stN = 0;
        }
        if (__ballot(m.stDfs) != 0ull) {
            const uint32_t o = chD.alloc(&q.cnt[5], dfsCap, m.stDfs ? 1u : 0u, 64u, ovD, holeD);
            if (m.stDfs && o != 0xFFFFFFFFu) {
                DfsTask t;
                t.rsId = m.rsId;
                t.scheme = (uint8_t)m.sel;
                t.search = (uint8_t)m.slot;
                t.idx = (uint8_t)m.stIdx;
                t.pad = 0;
                t.r = m.stR;
                t.depth = m.stDepth;
                dfsQ[o] = t;
            }
            m.stDfs = false;
        }
        if (__ballot(!done) == 0ull) break;
    }
    chI.fill(holeI);
    chD.fill(holeD);
    uint32_t flags = 0;
    if (ovI) flags |= FLAG_ITEM_OVERFLOW;
    if (ovD) flags |= FLAG_DFS_OVERFLOW;
    const uint32_t local[4] = {m.cNode, m.cExp, m.cImm, m.cStart};
    const int which[4] = {0, 7, 5, 6};
    flushCounters(q, local, which, 4);
    if (flags) atomicOr(&q.cnt[3], flags);
}

// ------------------------------------------------------------------ approximate search over the scheme
// (edit distance: the frontier kernels of dev_bfs_edit.hpp, included above)

// ------------------------------------------------------------------ locate + verification
constexpr int VROWS = MAX_READ + 3 * 7 + 4; // rows of the longest in-text matrix (len + Wv, Wv = 3 k, k <= 7)
__host__ __device__ inline uint32_t vRows(uint32_t maxLen) { return maxLen + 3u * 7u + 4u; } // ... of a batch

// Row storage of the traceback pass, interleaved by slot so that the lanes of a wavefront (which
// walk rows in lock step) write whole 512-byte lines: element (row, slot) lives at [row * nSlots + slot].
// What the traceback reads of row i is bit (j - 32 b) + DIAG of HP and D0 with |i - j| inside the band, i.e.
// at most 18 (= 3 k) bits below and 6 above the row's diagonal bit: both 32-bit windows starting 18 bits below
// the diagonal are kept in ONE 64-bit word per row (low half HP, high half D0).
// Layout: `lines` 64-byte lines per lane (slot), line-major (traceLine); line g = the packed rows 8 g + 1 .. 8 g + 8.  The forward pass
// collects eight rows in registers and writes the line with four 16-byte stores; the traceback fetches a line
// (and the eight text codes of its rows) whenever it crosses into the next group of eight rows.
struct VPlanes {
    uint64_t* W;
    uint32_t nSlots, lines;
};
// line g of a lane's trace rows.  Line-major: the 64 lines a wavefront writes (or reads) for one group of rows are
// 4 KB of consecutive memory instead of 64 lines `lines` x 64 bytes apart.
__device__ __forceinline__ uint4* traceLine(const VPlanes& V, uint32_t slot, uint32_t line) {
    return reinterpret_cast<uint4*>(V.W) + ((size_t)line * V.nSlots + slot) * 4;
}
constexpr uint32_t TB_BELOW = 18;  // narrow rows (32-bit matrix, maxED <= 4)
constexpr uint32_t TBW_BELOW = 21; // wide rows (64-bit / 16-row-block matrix, maxED <= 7: Wv = 3 maxED <= 21, Wh <= 7)
// low half: HP; high half: M | ~D0 — "the diagonal step is allowed" (bitparallelmatrix.h:559-562), folded in by the
// forward pass, which has the row's match word at hand: the trace then needs neither the text nor match words.
__device__ __forceinline__ uint64_t packTraceRow(uint32_t r, uint64_t HP, uint64_t diagOk) {
    const uint32_t sh = (r % MXW_BLOCK) + MXW_DIAG - TBW_BELOW;
    return (uint64_t)(uint32_t)(HP >> sh) | ((uint64_t)(uint32_t)(diagOk >> sh) << 32);
}
// NARROW rows (maxED <= 4): 16 + 16 bits per row, sixteen rows per 64-byte line — half the trace traffic.
// A trace only visits cells whose value is at most maxED, i.e. cells of the band j - i in [-3 maxED, maxED]
// (4 maxED + 1 <= 17 window bits, rel = j - i + TB_BELOW in [6, 22]); two of the 34 bits are never open:
//  * HP at the band's left edge (rel = 6): "horizontal" would come from a cell outside the band, whose value
//    exceeds maxED, so the bit is 0 wherever a trace reads it;
//  * "diagonal allowed" at the band's right edge (rel = 22): the alternative, a vertical step, would come from a
//    cell outside the band, so the bit is 1 wherever a trace (having read HP = 0) reads it.
// Kept: HP for rel 7..22 (low half), diagonal for rel 6..21 (high half).  The wide kernel CHECKS both rules on
// every step it takes (FLAG_CAPACITY if one is ever violated; CMB_TRACE_WIDE=1 selects it for any k).
constexpr uint32_t TBN_HP_LO = 7, TBN_DG_LO = 6, TBN_REL_LO = 6, TBN_REL_HI = 22, TBN_MAX_ED = 4;
// (HP and "diagonal allowed" of the 32-bit matrix: the diagonal of row r is bit r % 8 + MX32_DIAG)
__device__ __forceinline__ uint32_t packTraceRowNarrow(uint32_t r, uint32_t HP, uint32_t diagOk) {
    const uint32_t d = (r % MX32_BLOCK) + MX32_DIAG - TB_BELOW + 32u; // (+32: d itself would be negative)
    return ((HP >> (d + TBN_HP_LO - 32u)) & 0xFFFFu) | ((diagOk >> (d + TBN_DG_LO - 32u)) << 16);
}

// The device copy of the text holds CODES, one byte per character: A,C,G,T -> 0..3, anything else ('$',
// the padding behind the text) -> 4 (k_encode_text at index creation).
__global__ void k_encode_text(uint8_t* __restrict__ text, uint64_t n, uint64_t nPadded) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nPadded; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t ch = text[i];
        text[i] = i >= n ? 4 : ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
    }
}
__device__ __forceinline__ uint32_t textCode(uint8_t code) { return code; }
// The 2-bit copy of the text for the matrix kernels (a verification fetches 16 bytes per 64 rows instead of 64:
// one memory sector instead of two or three).  The rows a verification computes never reach the '$' (the window
// ends at n - 1, indexhelpers.cpp:527), so the copy is exact wherever it is used, PROVIDED the text has no other
// non-ACGT character: `bad` counts them and cmb_index_create then leaves DevIndex::text2 null (byte path).
__global__ void k_pack_text(const uint8_t* __restrict__ text, uint64_t n, uint64_t nWords, uint32_t* __restrict__ text2,
                            uint32_t* __restrict__ bad) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nWords; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(text)[w];
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
        uint32_t out = 0;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            const uint32_t c = (in[j >> 2] >> (8 * (j & 3u))) & 0xFFu;
            out |= (c & 3u) << (2 * j);
            if (c > 3u && 16 * w + j + 1 < n) atomicAdd(bad, 1u); // (position n - 1 is the '$')
        }
        text2[w] = out;
    }
}
// 32 characters from character position `pos` of the packed text: lo = characters 0..15, hi = 16..31
__device__ __forceinline__ void loadText2x32(const uint32_t* __restrict__ text2, uint32_t pos, uint32_t& lo, uint32_t& hi) {
    const uint32_t* p = text2 + (pos >> 4);
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
    const uint32_t sh = (pos & 15u) * 2u;
    lo = __funnelshift_r(w0, w1, sh);
    hi = __funnelshift_r(w1, w2, sh);
}

constexpr uint32_t ML_WORDS = 5 * 256; // LDS match-word table of a 256-thread block: [code 0..4][thread]

// The match words of the FULL read (the matrix of the in-text verification has the whole read as its horizontal
// sequence): for every read x strand and 32-row block the four 64-bit words, 32 bytes, computed once per batch
// (k_match_words) — the verification stages, the traceback's forward pass and the trace itself fetch them with
// two 16-byte loads instead of funnel-shifting twelve scattered 4-byte words of the bit-strings each time.
struct MFull {
    const uint4* p; // [read x strand][block][2]
    uint32_t nBlk;
};
__host__ __device__ inline uint32_t mfullBlocks(uint32_t maxLen) { return (maxLen + MXF_LEFT + 31u) / 32u + 1u; }
__global__ void k_match_words(const uint32_t* __restrict__ G, uint32_t gw, const uint64_t* __restrict__ offs, uint32_t nRs,
                              uint32_t nBlk, uint4* __restrict__ out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)nRs * nBlk) return;
    const uint32_t rs = (uint32_t)(gid / nBlk), b = (uint32_t)(gid % nBlk);
    const uint32_t len = (uint32_t)(offs[(rs >> 1) + 1] - offs[rs >> 1]);
    uint64_t m[4];
#pragma unroll
    for (int ch = 0; ch < 4; ch++) m[ch] = matchWord<MXF_LEFT>(gString(G, gw, rs, 0u, (uint32_t)ch), 0, len, b);
    out[gid * 2] = make_uint4((uint32_t)m[0], (uint32_t)(m[0] >> 32), (uint32_t)m[1], (uint32_t)(m[1] >> 32));
    out[gid * 2 + 1] = make_uint4((uint32_t)m[2], (uint32_t)(m[2] >> 32), (uint32_t)m[3], (uint32_t)(m[3] >> 32));
}
__device__ __forceinline__ void loadMatchWords(const MFull& mf, uint32_t rs, uint32_t b, uint64_t m[4]) {
    uint4 a = make_uint4(0, 0, 0, 0), c = a;
    if (b < mf.nBlk) {
        const uint4* q = mf.p + ((size_t)rs * mf.nBlk + b) * 2;
        a = q[0];
        c = q[1];
    }
    m[0] = (uint64_t)a.x | ((uint64_t)a.y << 32);
    m[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
    m[2] = (uint64_t)c.x | ((uint64_t)c.y << 32);
    m[3] = (uint64_t)c.z | ((uint64_t)c.w << 32);
}

// forward pass of the banded matrix of one candidate over the text window [start, start+size).
// STORE = false: verification pass (k_verify) — nothing is stored, cluster centres of the final column
//   are detected on the fly (bitparallelmatrix.h:591-614 needs only ED(i-1), ED(i), ED(i+1)).
// STORE = true : traceback pass (k_traceback) — HP and D0 of every row go to the interleaved planes.
// Returns the number of valid rows `i` (indexhelpers.cpp:535-539); centreMask bit t <=> row firstRow+1+t.
// CHECK = false (k_traceback: every row is known to be valid from pass 1): no rightmost-active-column walk, the
//   score comes from the row number and the count of diagonal matches; returns min(size, rows).
// rowMin: no lane has final-column rows (r > firstRow) at rows <= rowMin (wave-uniform skip of that code).
template <bool STORE, bool NARROW = false, bool PACKED = false, bool CHECK = true>
__device__ __forceinline__ uint32_t forwardPass(const DevIndex& ix, const MFull& mf, uint32_t rs,
                                                const MatGeom& g, uint32_t nZeros, uint32_t start, uint32_t size,
                                                uint32_t maxED, uint32_t minED, uint32_t& centreMask,
                                                uint64_t& edPack, uint64_t& edPackHi, const VPlanes& V, uint32_t slot,
                                                uint32_t& cRows, uint64_t* Ml, uint32_t rowMin = 0) {
    // NARROW (k <= 4) also means the matrix on 32-bit words / 8-row blocks (dev_matrix.hpp)
    typedef InTextMx<NARROW> MX; // NARROW: 32-bit words / 8-row blocks; else 64-bit words / 16-row blocks (dev_matrix.hpp)
    using W = typename MX::W;
    constexpr uint32_t LEFT = MX::LEFT, DIAG = MX::DIAG;
    W HP = (W)(~(W)0) << LEFT, HN = ((W)1 << (LEFT + 1u - nZeros)) - (W)1, D0 = 0, RAC = racInit((W)0, DIAG + g.Wh);
    uint32_t score = 0;
    const uint32_t sfc = g.sfc();
    const uint32_t firstRow = (g.m - 1) - sfc;
    const uint32_t col = g.n - 1;
    // sliding window of final-column values: edPrev = ED(r-1), edCur = ED(r) once r >= firstRow
    uint32_t edPrev2 = 0, edPrev = 0;
    if (!STORE && firstRow == 0) edPrev = MX::cell(0, col, HP, HN, score);
    centreMask = 0;
    edPack = 0; // 3 bits per final-column row above firstRow: min(ED, 7); rows 21.. go to edPackHi
    edPackHi = 0;
    // All lanes of the wavefront walk their rows in lock step (row r in iteration r - 1 for everybody), so the
    // loop counter, the byte position inside the 16-code text chunk and the moment the match words of the
    // next 32-row block are needed are wave-uniform; only "does this lane still have rows" is per lane.
    // The four match words of the lane's current block sit in LDS, indexed by the text code (code 4: zero).
    const uint32_t tid = threadIdx.x;
    Ml[4 * 256 + tid] = 0ull;
    const uint8_t* tp = ix.text + start;
    uint4 cur = make_uint4(0, 0, 0, 0), nxt = cur;
    // PACKED: the 16 characters of chunk c are a funnel shift of words c, c + 1 of the packed text from `start`
    const uint32_t* tp2 = PACKED ? ix.text2 + (start >> 4) : nullptr;
    const uint32_t sh2 = (start & 15u) * 2u;
    uint32_t pw0 = 0, pw1 = 0, pw2 = 0;
    if (PACKED) {
        pw0 = tp2[0];
        pw1 = tp2[1];
        pw2 = tp2[2];
    } else {
        cur = loadText16(tp);
        nxt = loadText16(tp + 16); // the text allocation is padded
    }
    uint32_t i = 0, dm = 0;
    W dAcc = 0; // CHECK = false: diagonal matches so far (dm) and of the current block of rows (one bit each)
    bool alive = size > 0;
    uint64_t buf[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // STORE: the packed rows of the current group of eight
    uint32_t bufN[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // NARROW: of the current group of sixteen
    bool groupAlive = false;
    for (uint32_t c = 0; __ballot(alive) != 0ull; c++) {
        const uint32_t pcur = PACKED ? __funnelshift_r(pw0, pw1, sh2) : 0u;
#pragma unroll
        for (uint32_t t = 0; t < 16; t++) {
            const uint32_t r = 16 * c + t + 1;
            if (STORE && (t & (NARROW ? 15u : 7u)) == 0) groupAlive = alive;
            if ((t == 15 && (c & 1u)) || (t == 0 && c == 0)) { // r % 32 == 0, or the first row: next block's words
                if (alive) {
                    uint64_t mw[4];
                    loadMatchWords(mf, rs, r / MX_BLOCK, mw);
#pragma unroll
                    for (int ch = 0; ch < 4; ch++) Ml[ch * 256 + tid] = mw[ch];
                }
            }
            const uint32_t wsel = t >> 2;
            const uint32_t wv = wsel == 0 ? cur.x : wsel == 1 ? cur.y : wsel == 2 ? cur.z : cur.w;
            const uint32_t tc = PACKED ? (pcur >> (2 * t)) & 3u : (wv >> (8 * (t & 3u))) & 0xFFu;
            if (alive) {
                const uint64_t M64 = Ml[tc * 256 + tid];
                const W M = MX::matchWordOf(M64, r);
                bool valid = true;
                if (CHECK) {
                    cRows++;
                    valid = MX::row(g, r, M, HP, HN, D0, RAC, score);
                } else {
                    MX::core(r, M, HP, HN, D0);
                    constexpr uint32_t BLOCK = MX::BLOCK;
                    dAcc |= D0 & ((W)1 << ((r % BLOCK) + DIAG)); // the diagonal cell matched (the bit moves with the row)
                    if ((r % BLOCK) == BLOCK - 1u) {
                        dm += MX::popc(dAcc);
                        dAcc = 0;
                    }
                }
                if (STORE && !NARROW) buf[t & 7u] = packTraceRow(r, (uint64_t)HP, (uint64_t)(M | ~D0));
                if (STORE && NARROW) bufN[t] = packTraceRowNarrow(r, (uint32_t)HP, (uint32_t)(M | ~D0));
                if (!valid) {
                    alive = false;
                } else if (!CHECK) {
                    if (STORE && r > rowMin && r > firstRow) {
                        const uint32_t ed = MX::cell(r, col, HP, HN, r - dm - MX::popc(dAcc));
                        const uint32_t bidx = r - firstRow - 1u;
                        if (bidx < 21u) edPack |= (uint64_t)min(ed, 7u) << (3u * bidx);
                        else edPackHi |= (uint64_t)min(ed, 7u) << (3u * (bidx - 21u));
                    }
                    i = r;
                    alive = r < size;
                } else {
                    if (STORE && r > firstRow) {
                        const uint32_t ed = MX::cell(r, col, HP, HN, score);
                        const uint32_t bidx = r - firstRow - 1u;
                        if (bidx < 21u) edPack |= (uint64_t)min(ed, 7u) << (3u * bidx);
                        else edPackHi |= (uint64_t)min(ed, 7u) << (3u * (bidx - 21u));
                    }
                    if (!STORE && r >= firstRow) {
                        const uint32_t ed = MX::cell(r, col, HP, HN, score);
                        // row r-1 can now be judged (its `below` neighbour is known)
                        if (r - 1 > firstRow) {
                            const uint32_t e1 = edPrev;
                            if (e1 <= maxED && e1 >= minED && e1 <= edPrev2 && e1 <= ed) centreMask |= 1u << (r - 2 - firstRow);
                        }
                        edPrev2 = edPrev;
                        edPrev = ed;
                    }
                    i++;
                    if (i >= size) alive = false;
                }
            }
            if (STORE && NARROW && t == 15u && groupAlive) { // rows 16 c + 1 .. 16 c + 16 = line c
                uint4* L = traceLine(V, slot, c);
#pragma unroll
                for (int h = 0; h < 4; h++) L[h] = make_uint4(bufN[4 * h], bufN[4 * h + 1], bufN[4 * h + 2], bufN[4 * h + 3]);
            }
            if (STORE && !NARROW && (t & 7u) == 7u && groupAlive) { // rows 16 c + t - 6 .. 16 c + t + 1 = line 2 c + t / 8
                uint4* L = traceLine(V, slot, 2 * c + (t >> 3));
#pragma unroll
                for (int h = 0; h < 4; h++)
                    L[h] = make_uint4((uint32_t)buf[2 * h], (uint32_t)(buf[2 * h] >> 32), (uint32_t)buf[2 * h + 1],
                                      (uint32_t)(buf[2 * h + 1] >> 32));
            }
        }
        if (PACKED) {
            pw0 = pw1;
            pw1 = pw2;
            pw2 = tp2[c + 3];
        } else {
            cur = nxt;
            nxt = loadText16(tp + 16 * (c + 2));
        }
    }
    if (!STORE && i > firstRow) { // the last valid row has no `below` neighbour (i == lastRow)
        const uint32_t e1 = edPrev;
        if (e1 <= maxED && e1 >= minED && e1 <= edPrev2) centreMask |= 1u << (i - 1 - firstRow);
    }
    return i;
}

// Key of an edit-distance verification: everything the banded matrix depends on.  The k+1 (or more) parts
// of one read that seed the same alignment produce the same key, so equal keys are verified once and every
// counter is scaled by the multiplicity (the reference verifies each of them, with identical results).
// Layout: read x strand (25 bits) | maxED | minED | fixed | start rotated right by VK_LOW.  Equal keys only have to
// end up NEXT TO each other (a missed merge costs a repeated verification, never a different result: the counters
// are scaled by the multiplicities), so the radix sort only covers the bits from the low VK_LOW bits of the start
// upwards — five 8-bit passes for a sub-batch of 2^22 reads instead of eight.
constexpr uint32_t VK_LOW = 10;
__host__ __device__ __forceinline__ unsigned long long packVerifyKey(uint32_t rs, uint32_t start, uint32_t maxED,
                                                                     uint32_t minED, uint32_t fixed) {
    return ((unsigned long long)rs << 39) | ((unsigned long long)(maxED & 7u) << 36) | ((unsigned long long)(minED & 7u) << 33) |
           ((unsigned long long)(fixed & 1u) << 32) | (unsigned long long)((start >> VK_LOW) | (start << (32u - VK_LOW)));
}
__host__ __device__ __forceinline__ uint32_t verifyKeyStart(unsigned long long key) {
    const uint32_t v = (uint32_t)key;
    return (v << VK_LOW) | (v >> (32u - VK_LOW));
}
// ... and for batches at 8 ... 13 errors (k_wide_filter / k_verify_wide): four bits for each bound, read x strand from bit 41
// (sub-batches of at most 2^20 reads)
constexpr uint32_t VKW_RS = 41;
__host__ __device__ __forceinline__ unsigned long long packVerifyKeyW(uint32_t rs, uint32_t start, uint32_t maxED, uint32_t minED,
                                                                      uint32_t fixed) {
    return ((unsigned long long)rs << VKW_RS) | ((unsigned long long)(maxED & 15u) << 37) | ((unsigned long long)(minED & 15u) << 33) |
           ((unsigned long long)(fixed & 1u) << 32) | (unsigned long long)((start >> VK_LOW) | (start << (32u - VK_LOW)));
}

// one edit-distance verification (FMIndex::inTextVerification + InTextVerificationTask::doTask) of `mult`
// identical candidates; returns true with a traceback task if the final column holds cluster centres
__device__ __forceinline__ bool verifyEdit(const DevIndex& ix, const uint64_t* offs, const MFull& mf,
                                           uint32_t rs, uint32_t start, uint32_t maxED, uint32_t minED, uint32_t fixed,
                                           uint32_t mult, uint32_t& cStarted, uint32_t& cRows, uint32_t& cText,
                                           uint32_t& cAbort, uint32_t& cCig, uint4& tbRec, uint64_t* Ml,
                                           uint32_t limitEnd = 0 /* explicit end of the text window (inTextVerificationOneString) */) {
    cStarted += mult;
    const uint32_t len = (uint32_t)(offs[(rs >> 1) + 1] - offs[rs >> 1]);
    const uint32_t nZeros = fixed ? 1u : 2u * maxED + 1u;
    MatGeom g;
    g.n = len + 1;
    g.maxED = maxED;
    g.Wv = nZeros - 1 + maxED;
    g.Wh = maxED;
    g.m = max(g.Wv + g.n, g.Wv + g.Wh + 1u); // (bitparallelmatrix.cpp:98-103: reads shorter than the band)
    const uint32_t maxEnd = ix.n - 1;
    const uint32_t hEnd = limitEnd ? min(maxEnd, limitEnd) : min(maxEnd, start + g.m - 1);
    const uint32_t size = hEnd > start ? hEnd - start : 0;
    if (!g.inFinalColumn(size)) return false;
    uint32_t mask = 0, rows = 0;
    uint64_t edPack, edPackHi;
    const VPlanes noPlanes{nullptr, 0, 0};
    const uint32_t i = forwardPass<false>(ix, mf, rs, g, nZeros, start, size, maxED, minED,
                                          mask, edPack, edPackHi, noPlanes, 0, rows, Ml);
    cRows += rows * mult;
    cText += rows * mult;
    if (i <= size - g.sfc() || mask == 0) { // indexhelpers.cpp:542, :550
        cAbort += mult;
        return false;
    }
    cCig += (uint32_t)__popc(mask) * mult; // = positions the traceback will report (one per centre)
    tbRec = make_uint4(rs, start, mask, maxED | (fixed << 4));
    return true;
}

// pass 1: locate + verify.  Exact / Hamming candidates produce text occurrences directly; edit-distance
// candidates whose final column holds cluster centres become traceback tasks {rs, start, mask, meta}.
// Four text codes (0..4) against four read codes (1..5, 5 = N): bit 8 b of the result is set iff character b
// differs — a text code 4 ('$', padding) never matches (verifyInTextExact / inTextVerificationHamming compare
// characters; '$' is not a read character).
__device__ __forceinline__ uint32_t mismatch4(uint32_t tw, uint32_t sw) {
    const uint32_t x = (tw + 0x01010101u) ^ sw;
    return ((x | (x >> 1) | (x >> 2)) | (tw >> 2)) & 0x01010101u;
}
// the same for a chunk of nC <= 16 characters: m[w] = flags of word w (bytes beyond nC cleared); returns the count
__device__ __forceinline__ uint32_t mismatch16(const uint4& t, const uint4& r, uint32_t nC, uint32_t m[4]) {
    const uint32_t tw[4] = {t.x, t.y, t.z, t.w}, sw[4] = {r.x, r.y, r.z, r.w};
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t nb = nC > 4 * w ? min(4u, nC - 4 * w) : 0u;
        const uint32_t valid = nb >= 4u ? 0x01010101u : (0x01010101u & ((1u << (8u * nb)) - 1u));
        m[w] = mismatch4(tw[w], sw[w]) & valid;
        cnt += (uint32_t)__popc(m[w]);
    }
    return cnt;
}

// KEYS: edit-distance candidates only get their verification key (the batch path; the matrix runs in
// k_verify_stage) — that instance carries no matrix code and keeps twice the wavefronts in flight for the locate.
template <bool KEYS>
__global__ void __launch_bounds__(256)
k_verify(DevIndex ix, const uint64_t* __restrict__ offs, uint32_t maxLen, uint32_t gw,
         const uint8_t* __restrict__ seq, MFull mf, const uint4* __restrict__ items,
         uint32_t nItems, uint4* __restrict__ tbq, uint32_t tbCap, unsigned long long* __restrict__ vkeys, Queues q,
         uint32_t wideKeys = 0 /* KEYS: the key layout of batches at 8 ... 13 errors (packVerifyKeyW) */) {
    __shared__ uint64_t Ml[ML_WORDS];
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cLF = 0, cLoc = 0, cText = 0, cRows = 0, cAbort = 0, cCig = 0, cStarted = 0, cRep = 0, flags = 0;
    WaveChunk chT, chB; // chunks of the text-occurrence queue / the traceback task queue
    bool ovT = false, ovB = false;
    auto holeT = [&](uint32_t i) { q.text[i].rsId = 0xFFFFFFFFu; };
    auto holeB = [&](uint32_t i) { tbq[i] = make_uint4(0, 0, 0, 0); }; // mask 0: nothing to trace
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t waveBase = slot & ~63u;
    const VPlanes noPlanes{nullptr, 0, 0};
    for (uint32_t base = waveBase; base < nItems; base += stride) { // wave-uniform trip count
        const uint32_t it = base + (threadIdx.x & 63u);
        uint32_t nOut = 0, nTb = 0;
        uint4 outRec = make_uint4(0, 0, 0, 0), tbRec = make_uint4(0, 0, 0, 0);
        uint32_t rs = 0;
        unsigned long long vkey = ~0ull;
        uint4 item = make_uint4(0xFFFFFFFFu, 0, 0, 0);
        if (it < nItems) item = items[it];
        if (item.x != 0xFFFFFFFFu) { // (holes: unused slots of a wavefront's chunk)
            rs = item.x;
            const uint32_t row = item.y, a = item.z, meta = item.w;
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
                    outRec = make_uint4(pos, pos + len, 0, 0);
                    nOut = 1;
                } else {
                    cStarted++;
                    const uint32_t remaining = a;
                    bool ok = pos >= remaining;
                    const uint32_t p0 = pos - remaining;
                    // 16 characters per step (the read codes sit in 16-byte aligned rows); the reference stops at the
                    // first mismatch: cText counts the characters up to and including it
                    for (uint32_t j = 0; ok && j < remaining; j += 16) {
                        const uint32_t nC = min(16u, remaining - j);
                        uint32_t m[4];
                        const uint32_t cnt = mismatch16(loadText16(ix.text + p0 + j), *reinterpret_cast<const uint4*>(s + j), nC, m);
                        if (cnt == 0) {
                            cText += nC;
                        } else {
                            const uint32_t w = m[0] ? 0u : m[1] ? 1u : m[2] ? 2u : 3u;
                            const uint32_t mw = w == 0 ? m[0] : w == 1 ? m[1] : w == 2 ? m[2] : m[3];
                            cText += 4 * w + (((uint32_t)__ffs(mw) - 1u) >> 3) + 1u;
                            ok = false;
                        }
                    }
                    if (ok) {
                        outRec = make_uint4(p0, p0 + len, 0, 0);
                        nOut = 1;
                    } else {
                        cAbort++;
                    }
                }
            } else if (kind == ITEM_HAMMING) { // FMIndex::inTextVerificationHamming (fmindex.cpp:370-406)
                cStarted++;
                const uint32_t lengthBefore = a;
                const uint32_t Tb = pos > lengthBefore ? pos - lengthBefore : 0;
                const uint32_t Te = Tb + len;
                if (Te <= ix.n) {
                    uint32_t score = 0;
                    for (uint32_t j = 0; j < len; j += 16) { // 16 characters per step
                        const uint32_t nC = min(16u, len - j);
                        uint32_t m[4];
                        const uint32_t cnt = mismatch16(loadText16(ix.text + Tb + j), *reinterpret_cast<const uint4*>(s + j), nC, m);
                        if (score + cnt <= maxED) {
                            score += cnt;
                            cText += nC;
                            continue;
                        }
                        // the reference breaks at the mismatch that exceeds maxED: find it (cText counts up to it)
                        for (uint32_t t = 0; t < nC; t++) {
                            const uint32_t mw = (t >> 2) == 0 ? m[0] : (t >> 2) == 1 ? m[1] : (t >> 2) == 2 ? m[2] : m[3];
                            cText++;
                            score += (mw >> (8u * (t & 3u))) & 1u;
                            if (score > maxED) break;
                        }
                        break;
                    }
                    if (score <= maxED && score >= minED) {
                        outRec = make_uint4(Tb, Te, score, 0);
                        nOut = 1;
                    }
                }
            } else {
                // ---- edit distance: FMIndex::inTextVerification + InTextVerificationTask::doTask
                const bool direct = (meta >> 23) & 1u; // (hooks: `a` is the explicit end of the window, or 0)
                const uint32_t startDiff = direct ? 0u : a;
                const uint32_t sum = pos + shift; // getBeginPositions (fmindex.h:374-379)
                const uint32_t start = sum >= startDiff ? sum - startDiff : 0;
                if (KEYS) { // verified once per distinct key by k_verify_stage
                    vkey = wideKeys ? packVerifyKeyW(rs, start, maxED, minED, fixed) : packVerifyKey(rs, start, maxED, minED, fixed);
                } else if (verifyEdit(ix, offs, mf, rs, start, maxED, minED, fixed, 1u, cStarted, cRows, cText, cAbort,
                                      cCig, tbRec, Ml, direct ? a : 0u)) {
                    nTb = 1;
                }
            }
        }
        if (KEYS && it < nItems) vkeys[it] = vkey;
        // ---- appends into the wavefront's chunks of the two output queues
        cRep += nOut;
        const uint32_t o1 = chT.alloc(&q.cnt[2], q.textCap, nOut, 256u, ovT, holeT);
        if (nOut && o1 != 0xFFFFFFFFu) q.text[o1] = TextOccRec{rs, outRec.x, outRec.y, outRec.z};
        const uint32_t o2 = chB.alloc(&q.cnt[7], tbCap, nTb, 256u, ovB, holeB);
        if (nTb && o2 != 0xFFFFFFFFu) tbq[o2] = tbRec;
    }
    chT.fill(holeT);
    chB.fill(holeB);
    if (ovT) flags |= FLAG_TEXT_OVERFLOW;
    if (ovB) flags |= FLAG_CAPACITY; // (sized by the host for the worst case)
    const uint32_t local[8] = {cLF, cLoc, cText, cRows, cAbort, cCig, cStarted, cRep + cCig};
    const int which[8] = {8, 9, 10, 11, 3, 4, 2, 1};
    flushCounters(q, local, which, 8);
    if (flags) atomicOr(&q.cnt[3], flags);
}

// In-text verification at 8 ... 13 errors (FMIndex::inTextVerification + InTextVerificationTask::doTask, fmindex.cpp:267-310,
// indexhelpers.cpp:518-574).  A candidate without a fixed start has a band of 4 k + 1 columns — 41 at k = 10 — where the reference switches
// to its 128-bit matrix; the SAME algorithm on 64-bit words with 16-row blocks and a left margin of 31 bits holds that band
// (dev_matrix.hpp: MXX_*), so these kernels run the reference's recurrence, rightmost-active-column walk, cluster centres and traceback
// as the staged path does for k <= 7, without its packed survivor / trace formats (which hold 7 errors):
//   k_wide_filter   which of the distinct verification keys reach their final column at all?  Most candidates are abandoned after a few
//                   dozen rows, the true locations run all len + 3 k of them.  A lane computes ONE row of its current candidate per
//                   turn of the loop and takes the next key as soon as its candidate has ended (a wavefront takes VW_WORK_CHUNK keys
//                   from the work counter at a time): the lanes stay busy whatever the lengths are; nothing is stored.  Candidates
//                   that fail the reference's abort test (indexhelpers.cpp:542) are counted here (started, rows, aborted — times
//                   their multiplicity); the others go on (`list`, per-wavefront chunks with holes).
//   k_verify_wide   computes those again in step, one candidate per lane, every row's {HP, M | ~D0} stored (16 bytes; the rows of a
//                   wavefront are contiguous), the values of the last column kept in registers; then findClusterCenters and traceBack
//                   (bitparallelmatrix.h:591-614, :531-586) on the stored rows.  KEYED: the batch path (keys, multiplicities, the
//                   list of k_wide_filter); otherwise the items of the hooks, each located and verified by itself.
constexpr uint32_t VW_ROW_BYTES = 16, VW_WORK_CHUNK = 1024;
__host__ __device__ inline uint32_t vwRows(uint32_t maxLen) { return maxLen + 3u * MXN_MAX_ED + 4u; }
// the two geometries of the wide in-text matrix (dev_matrix.hpp): 8 ... 10 errors, 11 ... 13 errors
struct WxTen {
    static constexpr uint32_t BLOCK = MXX_BLOCK, DIAG = MXX_DIAG, LEFT = MXX_LEFT;
};
struct WxThirteen {
    static constexpr uint32_t BLOCK = MXY_BLOCK, DIAG = MXY_DIAG, LEFT = MXY_LEFT;
};
template <class WX>
struct WideRow { // the matrix state of one candidate, with the text codes and match words of its current 16 rows
    static constexpr uint32_t NB = 16u / WX::BLOCK; // matrix blocks per 16 rows
    uint64_t HP, HN, RAC;
    uint32_t score;
    uint64_t Mw[NB][4]; // match words of the blocks, per text code
    uint4 tx;           // text codes of rows 16 c + 1 .. 16 c + 16
    uint32_t lastCode;  // ... and of row 16 c (the last one of the previous chunk)
    __device__ __forceinline__ void init(const MatGeom& g, uint32_t nZeros) { // (as forwardPass; bitparallelmatrix.cpp:105-121)
        HP = (~0ull) << WX::LEFT;
        HN = (1ull << (WX::LEFT + 1u - nZeros)) - 1ull;
        RAC = 1ull << (WX::DIAG + g.Wh);
        score = 0;
        tx = make_uint4(0, 0, 0, 0);
    }
    // before row 16 c (c = 0: before row 1): everything rows 16 c .. 16 c + 15 read from memory, in ONE round trip — a row by itself
    // would wait for its text code and then for the match word that code selects
    __device__ __forceinline__ void loadBlock(const DevIndex& ix, const uint32_t* G, uint32_t gw, uint32_t rs, uint32_t len, uint32_t start,
                                              uint32_t c) {
        lastCode = tx.w >> 24;
        tx = loadText16(ix.text + start + 16u * c); // (the text allocation is padded)
#pragma unroll
        for (uint32_t h = 0; h < NB; h++)
#pragma unroll
            for (uint32_t ch = 0; ch < 4; ch++) Mw[h][ch] = matchWord<WX::LEFT, WX::BLOCK>(gString(G, gw, rs, 0u, ch), 0u, len, c * NB + h);
    }
    // row r; M and D0 of the row are returned for the traceback
    __device__ __forceinline__ bool step(const MatGeom& g, uint32_t r, uint64_t& M, uint64_t& D0) {
        const uint32_t t = (r - 1u) & 15u, w = t >> 2;
        const uint32_t word = w == 0u ? tx.x : w == 1u ? tx.y : w == 2u ? tx.z : tx.w;
        const uint32_t tc = (r & 15u) == 0u ? lastCode : (word >> (8u * (t & 3u))) & 0xFFu; // (text code 4: '$' / padding, matches nothing)
        const uint32_t h = NB == 1u ? 0u : (r & 15u) / WX::BLOCK;
        uint64_t m0 = Mw[0][0], m1 = Mw[0][1], m2 = Mw[0][2], m3 = Mw[0][3];
        if (NB == 2u && h == 1u) m0 = Mw[NB - 1][0], m1 = Mw[NB - 1][1], m2 = Mw[NB - 1][2], m3 = Mw[NB - 1][3];
        M = tc == 0u ? m0 : tc == 1u ? m1 : tc == 2u ? m2 : tc == 3u ? m3 : 0ull;
        return computeRowWide<WX::BLOCK, WX::DIAG>(g, r, M, HP, HN, D0, RAC, score);
    }
};
__device__ __forceinline__ MatGeom wideGeom(uint32_t len, uint32_t maxED, uint32_t nZeros) {
    MatGeom g;
    g.n = len + 1u;
    g.maxED = maxED;
    g.Wv = nZeros - 1u + maxED;
    g.Wh = maxED;
    g.m = max(g.Wv + g.n, g.Wv + g.Wh + 1u); // (bitparallelmatrix.cpp:98-103: reads shorter than the band)
    return g;
}

template <class WX>
__global__ void __launch_bounds__(256)
k_wide_filter(DevIndex ix, const uint64_t* __restrict__ offs, const uint32_t* __restrict__ G, uint32_t gw,
              const unsigned long long* __restrict__ ukeys, const uint32_t* __restrict__ counts, uint32_t nKeys, uint32_t* __restrict__ work,
              uint32_t* __restrict__ list, uint32_t listCap, Queues q) {
    uint32_t cRows = 0, cAbort = 0, cStarted = 0, flags = 0;
    bool have = false, done = false;
    uint32_t u = 0, mult = 0, rs = 0, len = 0, r = 0, size = 0, start = 0, rows = 0;
    MatGeom g{0, 0, 0, 0, 0};
    WideRow<WX> mx;
    mx.HP = mx.HN = mx.RAC = 0, mx.score = 0, mx.tx = make_uint4(0, 0, 0, 0), mx.lastCode = 0;
    const uint32_t lane = threadIdx.x & 63u;
    // (atomics on one address are served at ~90 per microsecond, dev_wave.hpp: a chunk of keys per atomic, the survivors in
    // per-wavefront chunks of `list`, holes = 0xFFFFFFFF; work: [0] next key, [1] list slots handed out — both zero at launch)
    uint32_t chunkNext = 0, chunkEnd = 0; // (wave-uniform)
    WaveChunk chS;
    bool ovS = false;
    auto holeS = [&](uint32_t i) { list[i] = 0xFFFFFFFFu; };
    // The lanes of a wavefront keep their 16-row blocks in PHASE: a new candidate starts at a turn that is a multiple of 16 (a lane
    // waits 8 turns on average for it), so that all lanes load their next block — text codes, four match words — in the same turn,
    // one round trip per 16 rows of the whole wavefront.
    for (uint32_t turn = 0;; turn++) {
        bool skip = false;
        if ((turn & 15u) == 0u) { // (wave-uniform)
            bool fresh = false;
            const unsigned long long need = __ballot(!have && !done);
            if (need) {
                if (chunkNext == chunkEnd) {
                    uint32_t base = 0;
                    if (lane == 0u) base = atomicAdd(&work[0], VW_WORK_CHUNK);
                    chunkNext = (uint32_t)__shfl((int)base, 0);
                    chunkEnd = chunkNext + VW_WORK_CHUNK;
                }
                const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull)), avail = chunkEnd - chunkNext;
                const uint32_t idx = chunkNext + rank;
                chunkNext += min((uint32_t)__popcll(need), avail);
                if (!have && !done && rank < avail) { // (the others ask again 16 turns on: a new chunk)
                    if (idx >= nKeys) {
                        done = true;
                    } else {
                        const unsigned long long key = ukeys[idx];
                        mult = counts[idx];
                        if (key != ~0ull) { // (all ones: the run of the items that are no edit-distance candidates — it sorts last)
                            rs = (uint32_t)(key >> VKW_RS);
                            const uint32_t maxED = (uint32_t)(key >> 37) & 15u, fixed = (uint32_t)(key >> 32) & 1u;
                            const uint32_t nZeros = fixed ? 1u : 2u * maxED + 1u;
                            start = verifyKeyStart(key);
                            len = (uint32_t)(offs[(rs >> 1) + 1] - offs[rs >> 1]);
                            g = wideGeom(len, maxED, nZeros);
                            const uint32_t maxEnd = ix.n - 1, hEnd = min(maxEnd, start + g.m - 1);
                            size = hEnd > start ? hEnd - start : 0;
                            if (!g.inFinalColumn(size)) { // (indexhelpers.cpp:527): started, nothing computed
                                cStarted += mult;
                            } else if (g.Wv >= WX::LEFT) { // (the host picks the instance by the batch's distance)
                                flags |= FLAG_CAPACITY;
                            } else if (size == 0u) { // (no row to compute: the abort test of indexhelpers.cpp:542 with i = 0)
                                cStarted += mult;
                                cAbort += mult;
                            } else {
                                have = true;
                                fresh = true;
                                u = idx, r = 0u, rows = 0u;
                                mx.init(g, nZeros);
                            }
                        }
                    }
                }
            }
            if (!__any(have)) {
                if (__all(done)) break;
                turn = 0xFFFFFFFFu; // (nobody has work yet: ask again at once)
                continue;
            }
            if (have) mx.loadBlock(ix, G, gw, rs, len, start, r >> 4);
            skip = fresh; // (row 0 is the initial state: a fresh candidate's first row comes in the next turn)
            if (fresh) r = 1u;
        }
        bool survives = false;
        if (have && !skip) {
            uint64_t M, D0;
            const bool valid = mx.step(g, r, M, D0);
            rows++;
            if (!valid || r == size) {
                const uint32_t i = valid ? r : r - 1u; // the last valid row
                if (i <= size - g.sfc()) { // (length_t arithmetic as in the reference, indexhelpers.cpp:542)
                    cStarted += mult;
                    cRows += rows * mult;
                    cAbort += mult;
                } else {
                    survives = true;
                }
                have = false;
            } else {
                r++;
            }
        }
        if (__any(survives)) { // (wave-uniform)
            const uint32_t o = chS.alloc(&work[1], listCap, survives ? 1u : 0u, 256u, ovS, holeS);
            if (survives && o != 0xFFFFFFFFu) list[o] = u;
        }
    }
    chS.fill(holeS);
    if (ovS) flags |= FLAG_CAPACITY; // (sized by the host: every key and a chunk per wavefront)
    const uint32_t local[8] = {0u, 0u, cRows, cRows, cAbort, 0u, cStarted, 0u};
    const int which[8] = {8, 9, 10, 11, 3, 4, 2, 1};
    flushCounters(q, local, which, 8);
    if (flags) atomicOr(&q.cnt[3], flags);
}

template <bool KEYED, class WX>
__global__ void __launch_bounds__(256)
k_verify_wide(DevIndex ix, const uint64_t* __restrict__ offs, uint32_t maxLen, const uint8_t* __restrict__ seq, const uint32_t* __restrict__ G,
              uint32_t gw, const uint4* __restrict__ items, uint32_t nItems, const unsigned long long* __restrict__ ukeys,
              const uint32_t* __restrict__ counts, const uint32_t* __restrict__ list, const uint32_t* __restrict__ nList,
              uint8_t* __restrict__ slab, uint32_t slotBytes, Queues q) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x, nSlots = gridDim.x * blockDim.x, lane = threadIdx.x & 63u;
    // the slab of a wavefront: 16 bytes {HP, M | ~D0} per matrix row and lane, row-major — the lanes of a wavefront compute their rows in
    // step, so a row is ONE contiguous kilobyte
    uint4* const rowBits = reinterpret_cast<uint4*>(slab + (size_t)(slot - lane) * slotBytes) + lane;
    auto putRow = [&](uint32_t r, uint64_t a, uint64_t b) {
        rowBits[(size_t)r * 64u] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    };
    uint32_t cLF = 0, cLoc = 0, cRows = 0, cAbort = 0, cCig = 0, cStarted = 0, flags = 0;
    const uint32_t nIn = KEYED ? min(*nList, nItems) : nItems;
    WaveChunk chT; // the occurrences go into per-wavefront chunks of the text-occurrence queue (one atomic per 256 of them)
    bool ovT = false;
    auto holeT = [&](uint32_t o) { q.text[o].rsId = 0xFFFFFFFFu; };
    for (uint32_t base = slot - lane; base < nIn; base += nSlots) { // (wave-uniform trip count)
        const uint32_t it = base + lane;
        bool active = it < nIn;
        uint32_t rs = 0, maxED = 0, minED = 0, fixed = 0, start = 0, limitEnd = 0u, mult = 1u;
        if (active && KEYED) {
            const uint32_t u = list[it];
            if (u == 0xFFFFFFFFu) { // (holes of the list's per-wavefront chunks)
                active = false;
            } else {
                const unsigned long long key = ukeys[u];
                mult = counts[u];
                rs = (uint32_t)(key >> VKW_RS), maxED = (uint32_t)(key >> 37) & 15u, minED = (uint32_t)(key >> 33) & 15u, fixed = (uint32_t)(key >> 32) & 1u;
                start = verifyKeyStart(key);
            }
        } else if (active) {
            const uint4 item = items[it];
            const uint32_t meta = item.w;
            // (holes of the item queue; exact candidates of the k = 0 phases: k_verify)
            if (item.x == 0xFFFFFFFFu || ((meta >> 21) & 3u) != ITEM_EDIT) {
                active = false;
            } else {
                rs = item.x, maxED = (meta >> 12) & 15u, minED = (meta >> 16) & 15u, fixed = (meta >> 20) & 1u;
                const uint32_t shift = meta & 0xFFFu;
                const bool direct = (meta >> 23) & 1u; // (hooks: item.y is the start position itself, item.z the explicit end of the window or 0)
                uint32_t pos = item.y;
                if (!direct) {
                    cLoc++;
                    pos = findSA(ix, item.y, &cLF);
                }
                const uint32_t startDiff = direct ? 0u : item.z;
                limitEnd = direct ? item.z : 0u;
                const uint32_t sum = pos + shift; // getBeginPositions (fmindex.h:374-379)
                start = sum >= startDiff ? sum - startDiff : 0;
            }
        }
        const uint32_t len = active ? (uint32_t)(offs[(rs >> 1) + 1] - offs[rs >> 1]) : 0u;
        const uint8_t* rd = seq + (size_t)rs * maxLen;
        const uint32_t nZeros = fixed ? 1u : 2u * maxED + 1u;
        const MatGeom g = wideGeom(len, maxED, nZeros);
        const uint32_t sfc = g.sfc(), col = g.n - 1u, firstRow = (g.m - 1u) - sfc;
        // the values of the last column from row firstRow on (at most sfc + 1 <= 54 of them), four bits each: all that is asked of them
        // is how they compare with a value of at most maxED <= 13
        unsigned long long lcw0 = 0ull, lcw1 = 0ull, lcw2 = 0ull, lcw3 = 0ull;
        auto lcPut = [&](uint32_t qi, uint32_t v) {
            const unsigned long long x = (unsigned long long)min(v, 15u) << (4u * (qi & 15u));
            lcw0 |= (qi >> 4) == 0u ? x : 0ull, lcw1 |= (qi >> 4) == 1u ? x : 0ull, lcw2 |= (qi >> 4) == 2u ? x : 0ull, lcw3 |= (qi >> 4) == 3u ? x : 0ull;
        };
        auto lcGet = [&](uint32_t qi) -> uint32_t {
            const unsigned long long w = (qi >> 4) == 0u ? lcw0 : (qi >> 4) == 1u ? lcw1 : (qi >> 4) == 2u ? lcw2 : lcw3;
            return (uint32_t)(w >> (4u * (qi & 15u))) & 15u;
        };
        uint32_t i = 0;
        if (active) {
            cStarted += mult;
            const uint32_t maxEnd = ix.n - 1;
            const uint32_t hEnd = limitEnd ? min(maxEnd, limitEnd) : min(maxEnd, start + g.m - 1); // (limitEnd: inTextVerificationOneString)
            const uint32_t size = hEnd > start ? hEnd - start : 0;
            if (!g.inFinalColumn(size)) { // (indexhelpers.cpp:527)
                active = false;
            } else if (g.Wv >= WX::LEFT || (size + 1u) * VW_ROW_BYTES > slotBytes) {
                flags |= FLAG_CAPACITY;
                active = false;
            } else {
                WideRow<WX> mx;
                mx.init(g, nZeros);
                putRow(0u, mx.HP, ~0ull);
                if (firstRow == 0u) lcPut(0u, cellAt<WX::BLOCK, WX::DIAG>(0u, col, mx.HP, mx.HN, mx.score));
                for (uint32_t r = 1; r <= size; r++) {
                    if (r == 1u || (r & 15u) == 0u) mx.loadBlock(ix, G, gw, rs, len, start, r >> 4);
                    uint64_t M, D0;
                    const bool valid = mx.step(g, r, M, D0);
                    cRows += mult;
                    putRow(r, mx.HP, M | ~D0);
                    if (!valid) break;
                    if (r >= firstRow) lcPut(r - firstRow, cellAt<WX::BLOCK, WX::DIAG>(r, col, mx.HP, mx.HN, mx.score));
                    i = r;
                }
                if (i <= size - sfc) { // (length_t arithmetic as in the reference, indexhelpers.cpp:542)
                    cAbort += mult;
                    active = false;
                }
            }
        }
        // findClusterCenters (bitparallelmatrix.h:591-614), then traceBack (:531-586) of every centre: rows i, i - 1, ... firstRow + 1 —
        // the lanes of the wavefront walk their rows in step, so that the occurrences of a step are appended together
        const uint32_t nCand = active && i > firstRow ? i - firstRow : 0u;
        uint32_t nCentres = 0;
        for (uint32_t t = 0; __any(t < nCand); t++) {
            bool emit = false;
            TextOccRec rec{0xFFFFFFFFu, 0u, 0u, 0u};
            if (t < nCand) {
                const uint32_t r = i - t;
                const uint32_t ED = lcGet(r - firstRow); // (firstRow <= row <= i)
                if (ED <= maxED && ED >= minED) {
                    const bool betterThanAbove = r == firstRow || ED <= lcGet(r - 1u - firstRow);
                    const bool betterThanBelow = r == i || ED <= lcGet(r + 1u - firstRow);
                    if (betterThanAbove && betterThanBelow) {
                        nCentres++;
                        uint32_t ti = r, tj = col;
                        while (tj > 0) {
                            const uint32_t bitIdx = (tj - (ti / WX::BLOCK) * WX::BLOCK) + WX::DIAG; // (:541-543; unsigned as there)
                            if (bitIdx >= 64u) {
                                flags |= FLAG_CAPACITY; // (a path of cells <= maxED stays inside the band)
                                break;
                            }
                            const uint4 rb = rowBits[(size_t)ti * 64u];
                            const uint64_t hp = rb.x | ((uint64_t)rb.y << 32), md = rb.z | ((uint64_t)rb.w << 32);
                            if ((hp >> bitIdx) & 1ull) { // gap in horizontal (:553)
                                --tj;
                            } else if (ti > 0 && ((md >> bitIdx) & 1ull)) { // diagonal (:559): the characters match, or D0 is not set
                                --ti;
                                --tj;
                            } else if (ti > 0) { // gap in vertical
                                --ti;
                            } else {
                                flags |= FLAG_CAPACITY; // (row 0 only has horizontal steps)
                                break;
                            }
                        }
                        cCig += mult;
                        emit = true;
                        rec = TextOccRec{rs, start + ti, start + r, ED};
                    }
                }
            }
            if (__any(emit)) { // (wave-uniform)
                const uint32_t o = chT.alloc(&q.cnt[2], q.textCap, emit ? 1u : 0u, 256u, ovT, holeT);
                if (emit && o != 0xFFFFFFFFu) q.text[o] = rec;
            }
        }
        if (active && nCentres == 0) cAbort += mult; // indexhelpers.cpp:550
    }
    chT.fill(holeT);
    if (ovT) flags |= FLAG_TEXT_OVERFLOW;
    const uint32_t local[8] = {cLF, cLoc, cRows, cRows, cAbort, cCig, cStarted, cCig};
    const int which[8] = {8, 9, 10, 11, 3, 4, 2, 1};
    flushCounters(q, local, which, 8);
    if (flags) atomicOr(&q.cnt[3], flags);
}

// pass 1b: the distinct edit-distance verifications (sorted + run-length encoded keys of k_verify), in STAGES
// of nb 32-row matrix blocks each (nb = 2; the text below describes nb = 1).  FMIndex::inTextVerification + InTextVerificationTask::doTask
// (fmindex.cpp:267-310, indexhelpers.cpp:518-574) abandon most candidates after a few dozen rows while the true
// locations run all len + 3k rows; with one candidate per lane from start to end, a wavefront is as slow as its
// longest candidate (measured: 38 % of the lanes busy).  So stage s computes rows 32 s .. 32 s + 31 (stage 0:
// rows 1 .. 31) of every candidate that is still alive, in lock step, and appends the survivors — 36 bytes of
// row state each — to the work list of stage s + 1: every stage runs on a dense list.  A stage needs exactly two
// 16-byte text chunks and ONE set of match words (rows 32 s .. 32 s + 31 share matrix block s), fetched up front.
struct VStageList { // survivors entering a stage
    uint4* a;       // {key lo, key hi, multiplicity | score << 24, centre mask}
    uint4* b;       // {HP, HN}
    uint32_t* c;    // len | RAC bit << 16 | edPrev << 22 | edPrev2 << 27
};

// W32: the matrix on 32-bit words / 8-row blocks (dev_matrix.hpp; k <= 4) — half the VALU work of a row.
#ifdef CMB_STAGE_STATS
__device__ unsigned long long g_stageStats[24];
#endif
// FINALCOL: rows of this stage may lie in the final-column range of some candidate (row >= len - maxED - 1); the
// host clears it for the stages no read of the batch can reach that far in, and those instances carry no
// final-column / cluster-centre code (a third fewer instructions per row).
// Per row and alive lane the loop keeps no row counter: a lane that is alive has done every row up to the
// (wave-uniform) current one, so rows done, MATRIX_ROWS and the score (rows - matches on the diagonal) are derived
// from the row number; only the diagonal matches are counted.
template <bool FIRST, bool W32, bool PACKED, bool FINALCOL>
__global__ void __launch_bounds__(256)
k_verify_stage(DevIndex ix, const uint64_t* __restrict__ offs, MFull mf,
               const unsigned long long* __restrict__ ukeys, const uint32_t* __restrict__ counts, uint32_t nKeys,
               VStageList in, VStageList out, uint32_t* __restrict__ nList, uint32_t listCap, uint32_t stage,
               uint32_t nb, uint4* __restrict__ tbq, uint32_t tbCap, Queues q) {
    __shared__ uint64_t Ml[ML_WORDS];
    __shared__ uint32_t sh[2][5];
    const uint32_t tid = threadIdx.x;
    const uint32_t nIn = FIRST ? nKeys : min(nList[stage], listCap);
    uint32_t cText = 0, cAbort = 0, cCig = 0, cStarted = 0, flags = 0;
    Ml[4 * 256 + tid] = 0ull; // text code 4 ('$', padding): matches nothing
    const uint32_t rFirst = FIRST ? 1u : 32u * nb * stage; // first row of this stage (nb 32-row blocks per stage)
    const uint32_t rLast = 32u * nb * (stage + 1u) - 1u;   // last row of this stage
    for (uint32_t base = blockIdx.x * 256u; base < nIn; base += gridDim.x * 256u) { // block-uniform trip count
        const uint32_t it = base + tid;
        bool alive = false;
        unsigned long long key = ~0ull;
        uint32_t mult = 0, len = 0, score0 = 0, mask = 0, edPrev = 0, edPrev2 = 0;
        typedef InTextMx<W32> MX;
        using W = typename MX::W;
        constexpr uint32_t LEFT = MX::LEFT, DIAG = MX::DIAG, BLOCK = MX::BLOCK;
        W HP = 0, HN = 0, RAC = 0;
        if (it < nIn) {
            if (FIRST) {
                key = ukeys[it];
                if (key != ~0ull) { // (~0: the run of non-edit items and holes)
                    mult = min(counts[it], 0xFFFFFFu);
                    const uint32_t rs = (uint32_t)(key >> 39);
                    len = (uint32_t)(offs[(rs >> 1) + 1] - offs[rs >> 1]);
                    cStarted += mult;
                }
            } else {
                const uint4 ra = in.a[it], rb = in.b[it];
                const uint32_t rc = in.c[it];
                key = (unsigned long long)ra.x | ((unsigned long long)ra.y << 32);
                mult = ra.z & 0xFFFFFFu;
                score0 = ra.z >> 24;
                mask = ra.w;
                HP = W32 ? (W)rb.x : (W)((uint64_t)rb.x | ((uint64_t)rb.y << 32));
                HN = W32 ? (W)rb.z : (W)((uint64_t)rb.z | ((uint64_t)rb.w << 32));
                len = rc & 0xFFFFu;
                RAC = racInit((W)0, (rc >> 16) & 63u);
                edPrev = (rc >> 22) & 31u;
                edPrev2 = (rc >> 27) & 31u;
            }
        }
        const uint32_t rs = (uint32_t)(key >> 39), start = verifyKeyStart(key);
        const uint32_t maxED = (uint32_t)(key >> 36) & 7u, minED = (uint32_t)(key >> 33) & 7u, fixed = (uint32_t)(key >> 32) & 1u;
        const uint32_t nZeros = fixed ? 1u : 2u * maxED + 1u;
        MatGeom g;
        g.n = len + 1;
        g.maxED = maxED;
        g.Wv = nZeros - 1 + maxED;
        g.Wh = maxED;
        g.m = max(g.Wv + g.n, g.Wv + g.Wh + 1u); // (bitparallelmatrix.cpp:98-103: reads shorter than the band)
        const uint32_t sfc = g.sfc();
        const uint32_t firstRow = (g.m - 1) - sfc;
        const uint32_t col = g.n - 1;
        uint32_t size = 0;
        if (key != ~0ull) {
            const uint32_t maxEnd = ix.n - 1;
            const uint32_t hEnd = min(maxEnd, start + g.m - 1);
            size = hEnd > start ? hEnd - start : 0;
            alive = g.inFinalColumn(size); // indexhelpers.cpp:527 (candidates that cannot reach it do nothing)
            if (FIRST && alive) {
                HP = (W)(~(W)0) << LEFT;
                HN = ((W)1 << (LEFT + 1u - nZeros)) - (W)1; // first column: nZeros zeros, then 1, 2, ...
                RAC = racInit((W)0, DIAG + g.Wh);
                if (g.Wv > DIAG) flags |= FLAG_CAPACITY; // (the band must fit the matrix words: the host picks the kernel)
                if (firstRow == 0) {
                    if (FINALCOL) edPrev = MX::cell(0, col, HP, HN, 0);
                    else flags |= FLAG_CAPACITY; // (the host launches the final-column instance for such reads)
                }
            }
        }
        if (!FINALCOL && alive && rLast >= firstRow) flags |= FLAG_CAPACITY; // (host: stageNeedsFinalColumn)
        const bool alive0 = alive;
        uint32_t deadRow = 0;  // the row without a cell <= maxED that ended the candidate (it was computed: it counts)
        uint32_t dm = 0;       // rows of this stage whose diagonal cell matched (score = rows - dm)
        W dAcc = 0;            // ... of the current block of rows, one bit each
        for (uint32_t h = 0; h < nb; h++) {
        if (__ballot(alive) == 0ull) break; // (wave-uniform)
        const uint32_t blk = nb * stage + h;
        const bool head = FIRST && h == 0; // rows 1..31 of the matrix (there is no row 0 to compute)
        // rows r0 .. r0 + 31: t is the row's position in its 32-row (and 8-row) block in every block, so the shifts
        // of the block boundaries are compile-time decisions of the unrolled loop
        const uint32_t r0 = 32u * blk;
        uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0;
        uint32_t pLo = 0, pHi = 0;
        if (alive) {
            // the text character of row r is text[start + r - 1]; the head block loads from `start` and moves the
            // characters up by one (position start - 1 need not exist)
            if (PACKED) {
                loadText2x32(ix.text2, head ? start : start + (r0 - 1), pLo, pHi);
                if (head) {
                    pHi = (pHi << 2) | (pLo >> 30);
                    pLo <<= 2;
                }
            } else {
                const uint8_t* tp = head ? ix.text + start : ix.text + start + (r0 - 1);
                t0 = loadText16(tp);
                t1 = loadText16(tp + 16); // the text allocation is padded
                if (head) {
                    t1 = make_uint4(__funnelshift_l(t0.w, t1.x, 8), __funnelshift_l(t1.x, t1.y, 8), __funnelshift_l(t1.y, t1.z, 8),
                                    __funnelshift_l(t1.z, t1.w, 8));
                    t0 = make_uint4(t0.x << 8, __funnelshift_l(t0.x, t0.y, 8), __funnelshift_l(t0.y, t0.z, 8),
                                    __funnelshift_l(t0.z, t0.w, 8));
                }
            }
            uint64_t mw[4];
            loadMatchWords(mf, rs, blk, mw);
#pragma unroll
            for (int ch = 0; ch < 4; ch++) Ml[ch * 256 + tid] = mw[ch];
        }
#pragma unroll
        for (uint32_t t = 0; t < 32; t++) {
            const uint32_t r = r0 + t;
            const uint32_t wsel = (t >> 2) & 3u;
            const uint4 tw = t < 16 ? t0 : t1;
            const uint32_t wv = wsel == 0 ? tw.x : wsel == 1 ? tw.y : wsel == 2 ? tw.z : tw.w;
            const uint32_t tc = PACKED ? ((t < 16 ? pLo : pHi) >> (2 * (t & 15u))) & 3u : (wv >> (8 * (t & 3u))) & 0xFFu;
            if (alive && !(t == 0 && head)) { // (the head block has 31 rows)
                const uint64_t M64 = Ml[tc * 256 + tid];
                const W M = MX::matchWordOf(M64, r);
                W D0;
                MX::advance(r, RAC);
                MX::core(r, M, HP, HN, D0);
                bool valid = true;
#ifdef CMB_STAGE_STATS
                if (W32) { // what do the rows of a stage do?  [stage-class][wave rows, lane rows, any miss, lane misses, any slow, lane slow]
                    const bool miss = !racHit(D0, RAC), slow = miss && !(((uint32_t)HP >> (uint32_t)RAC) & 1u);
                    // rows whose diagonal cell exceeds maxED (only these need the RAC to know whether the row is valid)
                    const uint32_t scoreNow = score0 + (r - (rFirst - 1u)) - dm - (uint32_t)__popc((uint32_t)dAcc) -
                                              (uint32_t)(((uint32_t)D0 >> ((r % BLOCK) + DIAG)) & 1u);
                    const uint64_t am = __ballot(true), mm = __ballot(miss), sm = __ballot(slow), bm = __ballot(scoreNow > maxED);
                    if ((tid & 63u) == (uint32_t)__ffsll((unsigned long long)am) - 1u) {
                        unsigned long long* st = g_stageStats + 8 * (FIRST ? 0 : FINALCOL ? 2 : 1);
                        atomicAdd(&st[0], 1ull);
                        atomicAdd(&st[1], (unsigned long long)__popcll(am));
                        atomicAdd(&st[2], mm ? 1ull : 0ull);
                        atomicAdd(&st[3], (unsigned long long)__popcll(mm));
                        atomicAdd(&st[4], sm ? 1ull : 0ull);
                        atomicAdd(&st[5], (unsigned long long)__popcll(sm));
                        atomicAdd(&st[6], (unsigned long long)__popcll(bm));
                        atomicAdd(&st[7], bm ? 1ull : 0ull);
                    }
                }
#endif
                if (!racHit(D0, RAC)) { // (rare: what an end needs is recorded here, off the common path)
                    valid = MX::walk(g, r, HP, HN, RAC);
                    if (!valid) deadRow = r;
                }
                dAcc |= D0 & ((W)1 << ((r % BLOCK) + DIAG)); // the diagonal cell matched (the bit moves with the row)
                if ((r % BLOCK) == BLOCK - 1u) {
                    dm += MX::popc(dAcc);
                    dAcc = 0;
                }
                if (FINALCOL && valid && r >= firstRow) {
                    const uint32_t score = score0 + (r - (rFirst - 1u)) - dm - MX::popc(dAcc);
                    const uint32_t ed = min(MX::cell(r, col, HP, HN, score), 31u);
                    if (r - 1 > firstRow) { // row r-1 can now be judged (its `below` neighbour is known)
                        const uint32_t e1 = edPrev;
                        if (e1 <= maxED && e1 >= minED && e1 <= edPrev2 && e1 <= ed) mask |= 1u << (r - 2 - firstRow);
                    }
                    edPrev2 = edPrev;
                    edPrev = ed;
                }
                // rows done = r; the window ends in the final-column range (size >= m - sfc > firstRow: inFinalColumn above)
                alive = FINALCOL ? valid && r < size : valid;
            }
        }
        }
        dm += MX::popc(dAcc);
        const bool ended = alive0 && !alive;
        // rows done so far: the stage's, all `size` of them, or those before the invalid row
        const uint32_t i = !ended ? rLast : deadRow ? deadRow - 1u : max(size, rFirst);
        const uint32_t rows = alive0 ? i - (rFirst - 1u) + (deadRow ? 1u : 0u) : 0u;
        const uint32_t score = score0 + rows - dm;
        cText += rows * mult;
        uint32_t nTb = 0;
        uint4 tbRec = make_uint4(0, 0, 0, 0);
        if (ended) {
            if (FINALCOL && i > firstRow) { // the last valid row has no `below` neighbour
                const uint32_t e1 = edPrev;
                if (e1 <= maxED && e1 >= minED && e1 <= edPrev2) mask |= 1u << (i - 1 - firstRow);
            }
            if (i <= size - sfc || mask == 0) { // indexhelpers.cpp:542, :550
                cAbort += mult;
            } else {
                cCig += (uint32_t)__popc(mask) * mult; // = positions the traceback will report
                tbRec = make_uint4(rs, start, mask, maxED | (fixed << 4));
                nTb = 1;
            }
        }
        uint32_t tA, tB;
        const uint32_t oS = blockAppend(&nList[stage + 1], alive ? 1u : 0u, sh[0], tA);
        const uint32_t oT = blockAppend(&q.cnt[7], nTb, sh[1], tB);
        if (alive) {
            if (oS >= listCap) flags |= FLAG_CAPACITY; // (sized for every candidate)
            else {
                if (score > 255u) flags |= FLAG_CAPACITY;
                out.a[oS] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), mult | (score << 24), mask);
                out.b[oS] = make_uint4((uint32_t)HP, (uint32_t)((uint64_t)HP >> 32), (uint32_t)HN, (uint32_t)((uint64_t)HN >> 32));
                out.c[oS] = len | (racIndex(RAC) << 16) | (edPrev << 22) | (edPrev2 << 27);
            }
        }
        if (nTb) {
            if (oT >= tbCap) flags |= FLAG_CAPACITY;
            else tbq[oT] = tbRec;
        }
    }
    const uint32_t local[6] = {cText, cText, cAbort, cCig, cStarted, cCig};
    const int which[6] = {10, 11, 3, 4, 2, 1}; // (every centre is reported once: TOTAL_REPORTED += CIGARS)
    flushCounters(q, local, which, 6);
    if (flags) atomicOr(&q.cnt[3], flags);
}

// pass 2: traceback of the candidates that hold cluster centres (about a third of them): the rows are
// recomputed, this time keeping HP and D0, and every centre is traced back to its begin row
// (bitparallelmatrix.h:531-586).  The traceback reads rows through an 8-row window staged in LDS, so
// the dependent chain costs one memory round trip per 8 rows instead of three per row.
constexpr int TBW = 8;
template <bool NARROW, bool PACKED>
__global__ void __launch_bounds__(256)
k_traceback(DevIndex ix, const uint64_t* __restrict__ offs, MFull mf,
            const uint4* __restrict__ tbq, uint32_t nTasks, VPlanes V, Queues q, uint32_t rowMin) {
    __shared__ uint64_t wW[NARROW ? 1 : TBW][256];
    __shared__ uint64_t Ml[ML_WORDS];
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tid = threadIdx.x;
    uint32_t flags = 0, dummyRows = 0;
    WaveChunk chT; // chunk of the text-occurrence queue
    bool ovT = false;
    auto holeT = [&](uint32_t i) { q.text[i].rsId = 0xFFFFFFFFu; };
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t waveBase = slot & ~63u;
    const uint64_t HP0 = (~0ull) << MXW_LEFT;
    for (uint32_t base = waveBase; base < nTasks; base += stride) { // wave-uniform trip count
        const uint32_t it = base + (tid & 63u);
        uint32_t rs = 0, start = 0, m = 0, firstRow = 0, len = 0, col = 0;
        uint32_t relLeft = 0, relRight = 0; // the band's edges in window bits
        uint64_t edPack = 0, edPackHi = 0;
        uint4 t = make_uint4(0, 0, 0, 0);
        if (it < nTasks) t = tbq[it];
        if (t.z != 0u) { // (mask 0: a hole of the task queue)
            rs = t.x;
            start = t.y;
            m = t.z;
            const uint32_t maxED = t.w & 15u, fixed = (t.w >> 4) & 1u;
            len = (uint32_t)(offs[(rs >> 1) + 1] - offs[rs >> 1]);
            const uint32_t nZeros = fixed ? 1u : 2u * maxED + 1u;
            MatGeom g;
            g.n = len + 1;
            g.maxED = maxED;
            g.Wv = nZeros - 1 + maxED;
            g.Wh = maxED;
            g.m = max(g.Wv + g.n, g.Wv + g.Wh + 1u); // (bitparallelmatrix.cpp:98-103: reads shorter than the band)
            firstRow = (g.m - 1) - g.sfc();
            col = g.n - 1;
            relLeft = TBW_BELOW - g.Wv; // (the wide rows' window; used by the wide walk only)
            relRight = TBW_BELOW + g.Wh;
            if (!NARROW && (g.Wv > MXW_DIAG || maxED > MXW_MAX_ED)) { // the band must fit the 64-bit / 16-row-block matrix
                flags |= FLAG_CAPACITY;
                m = 0;
            }
            if (NARROW && maxED > TBN_MAX_ED) { // (the host picks the wide kernel for k > 4)
                flags |= FLAG_CAPACITY;
                m = 0;
            }
            const uint32_t topCentre = firstRow + 1 + (31u - (uint32_t)__clz(m));
            uint32_t dummyMask;
            // rows 1..topCentre (all valid: they were valid in pass 1)
            forwardPass<true, NARROW, PACKED, false>(ix, mf, rs, g, nZeros, start, topCentre, maxED, 0, dummyMask, edPack,
                                                     edPackHi, V, slot, dummyRows, Ml, rowMin);
        }
        // one centre per lane and round; the wavefront appends its results with one atomic per round
        for (;;) {
            const bool have = m != 0;
            if (__ballot(have) == 0ull) break;
            TextOccRec rec{0, 0, 0, 0};
            uint32_t bitIdx = 0, ri = 0, ti = 0, tj = 0;
            if (have) {
                bitIdx = 31u - (uint32_t)__clz(m);
                m &= ~(1u << bitIdx);
                ri = firstRow + 1 + bitIdx;
                ti = ri;
                tj = col;
            }
            if (NARROW) {
                // The wavefront walks its traces ROW by row, downwards and in lock step (line g = rows 16 g + 1 ..
                // 16 g + 16 sits in sixteen registers, row j of it is handled by every lane whose trace is in that
                // row): per row, the run of horizontal steps is the run of HP bits below the current column (one
                // count-leading-ones), then one diagonal or vertical step — no loop over steps, no LDS.
                bool done = !have;
                uint32_t rel = tj + TB_BELOW - ti; // bit of the row's windows
                uint32_t gTop = have ? (ti - 1u) >> 4 : 0u;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) gTop = max(gTop, (uint32_t)__shfl_xor((int)gTop, d));
                for (int gq = (int)gTop; gq >= 0; gq--) { // (wave-uniform)
                    if (__ballot(!done) == 0ull) break;
                    uint32_t w[16];
#pragma unroll
                    for (int h = 0; h < 16; h++) w[h] = 0;
                    if (!done && ((ti - 1u) >> 4) == (uint32_t)gq) { // (a trace that has reached row 0 loads nothing)
#pragma unroll
                        for (int h = 0; h < 4; h++) {
                            const uint4 v = traceLine(V, slot, (uint32_t)gq)[h];
                            w[4 * h] = v.x;
                            w[4 * h + 1] = v.y;
                            w[4 * h + 2] = v.z;
                            w[4 * h + 3] = v.w;
                        }
                    }
#pragma unroll
                    for (int j = 15; j >= 0; j--) {
                        if (!done && ti == 16u * (uint32_t)gq + (uint32_t)j + 1u) {
                            const uint32_t wn = w[j];
                            if (rel - TBN_REL_LO > TBN_REL_HI - TBN_REL_LO) { // outside the band (checked, not assumed)
                                flags |= FLAG_CAPACITY;
                                done = true;
                            } else {
                                // HP of column rel is bit rel - TBN_HP_LO of the low half: move it to bit 31 and count the ones
                                const uint32_t y = (wn << 16) << (TBN_REL_HI - rel);
                                const uint32_t run = min((uint32_t)__clz(~y), tj); // gaps in horizontal (:553)
                                tj -= run;
                                rel -= run;
                                if (tj == 0) {
                                    done = true;
                                } else {
                                    // "diagonal allowed" of column rel is bit 16 + rel - TBN_DG_LO; always at the band's right edge
                                    const uint32_t z = __funnelshift_r(wn, 1u, rel - TBN_DG_LO);
                                    const uint32_t dg = (z >> 16) & 1u; // diagonal (:559); else vertical
                                    tj -= dg;
                                    rel += 1u - dg;
                                    ti -= 1u;
                                    if (tj == 0) done = true;
                                }
                            }
                        }
                    }
                }
                if (!done) { // row 0: gaps in horizontal down to column 0
                    if (rel > TBN_REL_HI) flags |= FLAG_CAPACITY;
                    tj = 0;
                }
            } else if (have) {
                uint32_t curG = 0xFFFFFFFFu; // the line (rows 8 g + 1 .. 8 g + 8) held in wW[.][tid]
                while (tj > 0) {
                    const uint32_t rel = tj + TBW_BELOW - ti; // bit of the row's windows
                    uint64_t ww = packTraceRow(0, HP0, 0ull); // row 0 (never steps diagonally: ti > 0 below)
                    if (ti > 0) {
                        const uint32_t gq = (ti - 1) >> 3, jq = (ti - 1) & 7u;
                        if (gq != curG) {
                            curG = gq;
                            const uint4* L = traceLine(V, slot, gq);
#pragma unroll
                            for (int h = 0; h < 4; h++) {
                                const uint4 v = L[h];
                                wW[2 * h][tid] = (uint64_t)v.x | ((uint64_t)v.y << 32);
                                wW[2 * h + 1][tid] = (uint64_t)v.z | ((uint64_t)v.w << 32);
                            }
                        }
                        ww = wW[jq][tid];
                    }
                    if (rel > 31u) { // outside the stored window: cannot happen inside the band (checked, not assumed)
                        flags |= FLAG_CAPACITY;
                        break;
                    }
                    const bool hpBit = ((uint32_t)ww >> rel) & 1u;
                    const bool dgBit = ((uint32_t)(ww >> 32) >> rel) & 1u;
                    // the two rules the narrow rows rely on (see packTraceRowNarrow), checked on every step
                    if (rel < relLeft || rel > relRight || (rel == relLeft && hpBit) ||
                        (rel == relRight && ti > 0 && !hpBit && !dgBit))
                        flags |= FLAG_TRACE_RULE;
                    if (hpBit) { // gap in horizontal (:553)
                        --tj;
                    } else {
                        if (ti > 0 && dgBit) --tj; // diagonal (:559); else vertical
                        --ti;
                    }
                }
            }
            if (have) {
                const uint32_t ed = bitIdx < 21u ? (uint32_t)((edPack >> (3u * bitIdx)) & 7ull)
                                                 : (uint32_t)((edPackHi >> (3u * (bitIdx - 21u))) & 7ull);
                rec = TextOccRec{rs, start + ti, start + ri, ed};
            }
            const uint32_t o = chT.alloc(&q.cnt[2], q.textCap, have ? 1u : 0u, 256u, ovT, holeT);
            if (have && o != 0xFFFFFFFFu) q.text[o] = rec;
        }
    }
    chT.fill(holeT);
    if (ovT) flags |= FLAG_TEXT_OVERFLOW;
    if (flags) atomicOr(&q.cnt[3], flags);
}

// Occurrences::eraseDoublesFM (indexhelpers.h:2135-2146) on the device: the in-index occurrences are sorted
// by (read, begin of the SA range) — one 64-bit radix sort — and a record is dropped if an identical one
// (FMOcc::operator==, :1529: same strand, range, distance, depth, shift) precedes it in its (short) run of
// equal keys.  The survivors' SA rows are what getUniqueTextOccurrences reports (:1378, :1390).
__global__ void k_fm_keys(const FMOccRec* __restrict__ fm, uint32_t n, unsigned long long* __restrict__ keys,
                          uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FMOccRec f = fm[i];
    keys[i] = f.rsId == 0xFFFFFFFFu ? ~0ull : (((unsigned long long)(f.rsId >> 1) << 32) | f.b); // holes last
    idx[i] = i;
}
__global__ void __launch_bounds__(256)
k_fm_unique(const FMOccRec* __restrict__ fm, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ idx,
            uint32_t n, FMOccRec* __restrict__ out, uint32_t* __restrict__ nOut, Queues q) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    FMOccRec f{};
    if (j < n && keys[j] != ~0ull) {
        f = fm[idx[j]];
        keep = true;
        for (uint32_t t = j; t-- > 0 && keys[t] == keys[j];) {
            const FMOccRec g = fm[idx[t]];
            if (g.rsId == f.rsId && g.e == f.e && g.dist == f.dist && g.depth == f.depth && g.shift == f.shift) {
                keep = false;
                break;
            }
        }
    }
    uint32_t total;
    const uint32_t o = waveAppend(nOut, keep ? 1u : 0u, total);
    if (keep) out[o] = f;
    unsigned long long rows = keep ? (unsigned long long)(f.e - f.b) : 0ull;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) rows += __shfl_xor(rows, d);
    if ((threadIdx.x & 63u) == 0 && rows) atomicAdd(&q.counters[1], rows); // TOTAL_REPORTED_POSITIONS
}

// in-index occurrences (already de-duplicated per read) -> text occurrences.  A wavefront reserves the slots of
// all the SA rows of its 64 occurrences with ONE atomic (prefix sum over the range widths).
__global__ void __launch_bounds__(256)
k_fmocc(DevIndex ix, const FMOccRec* __restrict__ recs, uint32_t n, Queues q) {
    uint32_t cLF = 0, cLoc = 0, flags = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t first = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t base = first & ~63u; base < n; base += stride) { // wave-uniform trip count
        const uint32_t i = base + (threadIdx.x & 63u);
        FMOccRec f{};
        uint32_t w = 0;
        if (i < n) {
            f = recs[i];
            w = f.e - f.b;
        }
        uint32_t total;
        const uint32_t o = waveAppend(&q.cnt[2], w, total);
        if (o + w > q.textCap) {
            if (w) flags |= FLAG_TEXT_OVERFLOW;
            continue;
        }
        for (uint32_t t = 0; t < w; t++) {
            cLoc++;
            const uint32_t p = findSA(ix, f.b + t, &cLF) + f.shift;
            q.text[o + t] = TextOccRec{f.rsId, p, p + f.depth, f.dist};
        }
    }
    const uint32_t local[2] = {cLF, cLoc};
    const int which[2] = {8, 9};
    flushCounters(q, local, which, 2);
    if (flags) atomicOr(&q.cnt[3], flags);
}

// ------------------------------------------------------------------ occurrence filter
// IndexInterface::getUniqueTextOccurrences / getTextOccHamming post-processing (reference
// src/indexinterface.cpp:1331-1491) on the device.  Every raw text occurrence is packed into ONE 64-bit
// key whose natural order is the reference's TextOcc::operator< (src/indexhelpers.h:779-795) within a
// read:   read[63:40] | begin[39:8] | distance[7:5] | (width - (len - k))[4:1] | strand[0]
// Hamming distance (every occurrence has the width of the read): distance[7:4] | 0[3:1] | strand[0] — four bits, for up to 13 errors
// One radix sort of the keys (rocPRIM) therefore groups the occurrences per read AND orders them;
// k_filter then walks each read's segment once: unique (same range and distance, :811), then the
// redundancy filter (:1447-1485).
__global__ void __launch_bounds__(256)
k_pack_keys(const TextOccRec* __restrict__ text, uint32_t n, const uint64_t* __restrict__ offs, uint32_t k,
            unsigned long long* __restrict__ keys, uint32_t* __restrict__ cnt,
            uint32_t perStrand /* BEST mode filters every strand by itself (mapRead, searchstrategy.h:490-523): the group of a
                                  key is then read x strand, not the read */,
            const uint8_t* __restrict__ only = nullptr /* dev_bfs_naive.hpp: keys for the reads marked here (bit 7), holes for the rest */,
            uint32_t layout = 0 /* keyBits */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TextOccRec t = text[i];
    if (t.rsId == 0xFFFFFFFFu || (only && !(only[t.rsId] & 0x80u))) { // a hole: sorts behind every read
        keys[i] = ~0ull;
        return;
    }
    const uint32_t r = t.rsId >> 1;
    const uint32_t len = (uint32_t)(offs[r + 1] - offs[r]);
    const uint32_t width = t.end - t.begin;
    const uint32_t wrel = width - (len - k); // in [0, 2k] for every occurrence of a read of length len
    const uint32_t grp = perStrand ? t.rsId : r;
    const KeyBits kb = keyBits(layout);
    uint32_t low;
    if (layout == 1u) {
        if (width != len || t.dist > kb.distMask) atomicOr(&cnt[3], (uint32_t)FLAG_CAPACITY);
        low = (t.dist & kb.distMask) << kb.dist;
    } else {
        if (wrel > kb.wMask || t.dist > kb.distMask) atomicOr(&cnt[3], (uint32_t)FLAG_CAPACITY);
        low = ((t.dist & kb.distMask) << kb.dist) | ((wrel & kb.wMask) << 1);
    }
    if ((unsigned long long)grp >> (64u - kb.group)) atomicOr(&cnt[3], (uint32_t)FLAG_CAPACITY);
    keys[i] = ((unsigned long long)grp << kb.group) | ((unsigned long long)t.begin << kb.begin) | (unsigned long long)low |
              (unsigned long long)(perStrand ? 0u : (t.rsId & 1u));
}

// segment of every read in the sorted keys: segBeg[r] = first key of read r (one coalesced pass; reads without
// occurrences keep the 0xFFFFFFFF the array was filled with and are skipped by k_filter)
__global__ void k_filter_segments(const unsigned long long* __restrict__ keys, uint32_t n, uint32_t* __restrict__ segBeg,
                                  uint32_t* __restrict__ segEnd, uint32_t groupShift = 40u) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = keys[i];
    if (key == ~0ull) return; // holes sort behind every read
    const uint32_t r = (uint32_t)(key >> groupShift);
    if (i == 0 || (uint32_t)(keys[i - 1] >> groupShift) != r) segBeg[r] = i;
    if (i + 1 == n || keys[i + 1] == ~0ull || (uint32_t)(keys[i + 1] >> groupShift) != r) segEnd[r] = i + 1;
}

// mode 0: k = 0 (no filtering, searchstrategy.cpp:499-510); 1: Hamming (unique only); 2: edit distance
// k_filter_mark: one lane per read walks the read's (short) segment of the sorted keys — the redundancy filter
// (:1447-1485) is sequential in the last occurrence it kept — and gives every key of the segment its rank among
// the read's surviving occurrences, or NONE.  Segments of up to FILTER_SHORT keys are fetched in one go by their
// lane; the few long ones (reads inside repeats: thousands of occurrences — one lane walking them alone set the
// duration of the whole kernel) are walked by their lane while the WAVEFRONT fetches them, 64 keys per load.
// k_filter_write: one lane per KEY (coalesced) writes the survivors at offset(read) + rank.
constexpr uint32_t FILTER_NONE = 0xFFFFFFFFu;
constexpr uint32_t FILTER_SHORT = 24;
__global__ void __launch_bounds__(256)
k_filter_mark(const unsigned long long* __restrict__ keys, uint32_t nReads, uint32_t k, int mode,
              uint32_t* __restrict__ counts, uint32_t* __restrict__ rank, const uint32_t* __restrict__ segBeg,
              const uint32_t* __restrict__ segEnd, uint32_t layout = 0u) {
    const KeyBits kb = keyBits(layout);
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t segLo = 0xFFFFFFFFu;
    if (r < nReads) segLo = segBeg[r];
    const bool has = segLo != 0xFFFFFFFFu; // (no occurrence of this read otherwise)
    const uint32_t nSeg = has ? segEnd[r] - segLo : 0u;
    const uint32_t maxDiff = 2 * k;
    // the filter's state and one step of it: returns the key's rank among the survivors (or NONE) and, in
    // `replaced`, the index of the previously kept key if this one takes its place (else NONE)
    struct State {
        uint32_t nOut = 0, lastKept = 0, prevBegin = 0xFFFFFFFFu, prevDepth = 0xFFFFFFFFu, prevED;
        unsigned long long prevKey = ~0ull;
    };
    auto stepCore = [&](State& st, uint32_t i, unsigned long long key, uint32_t& replaced) -> uint32_t {
        const uint32_t begin = (uint32_t)(key >> kb.begin), dist = (uint32_t)(key >> kb.dist) & kb.distMask; // (mode 2; the other modes compare whole keys)
        const uint32_t width = (uint32_t)(key >> 1) & kb.wMask; // relative to len - k: the same for all keys of a read
        replaced = FILTER_NONE;
        bool keep = true;
        if (mode != 0) {
            if ((key >> 1) == (st.prevKey >> 1)) keep = false; // same range and distance (either strand)
            else st.prevKey = key;
        }
        if (keep && mode == 2) {
            const uint32_t diff = begin > st.prevBegin ? begin - st.prevBegin : st.prevBegin - begin;
            if (diff == 0) keep = false;
            else if (diff <= maxDiff) {
                if (dist > st.prevED || (dist == st.prevED && width >= st.prevDepth)) keep = false;
                else {
                    st.nOut--; // the previous one was worse: replace it
                    replaced = st.lastKept;
                }
            }
            if (keep) {
                st.prevBegin = begin;
                st.prevED = dist;
                st.prevDepth = width;
            }
        }
        uint32_t mine = FILTER_NONE;
        if (keep) {
            mine = st.nOut++;
            st.lastKept = i;
        }
        return mine;
    };
    State mySt;
    mySt.prevED = k + 1;
    auto step = [&](uint32_t i, unsigned long long key) { // a lane walking its own short segment
        uint32_t replaced;
        const uint32_t mine = stepCore(mySt, i, key, replaced);
        if (replaced != FILTER_NONE) rank[segLo + replaced] = FILTER_NONE;
        rank[segLo + i] = mine;
    };
    const bool isShort = nSeg <= FILTER_SHORT;
    if (has && isShort) {
        unsigned long long first[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) first[j] = j < nSeg ? keys[segLo + j] : 0ull;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
            if (j < nSeg) step(j, first[j]);
        if (nSeg > 8u) {
            unsigned long long nxt[FILTER_SHORT - 8];
#pragma unroll
            for (uint32_t j = 0; j < FILTER_SHORT - 8; j++) nxt[j] = 8u + j < nSeg ? keys[segLo + 8u + j] : 0ull;
#pragma unroll
            for (uint32_t j = 0; j < FILTER_SHORT - 8; j++)
                if (8u + j < nSeg) step(8u + j, nxt[j]);
        }
    }
    // Long segments: the wavefront fetches 64 keys per load (the next 64 are on their way meanwhile) and ALL lanes
    // run the filter on them with the same, wave-uniform state (scalar code: the key of step j comes from lane j by
    // v_readlane); lane j keeps the rank of key j and the 64 ranks leave in one coalesced store.
    unsigned long long todo = __ballot(has && !isShort);
    while (todo) { // wave-uniform
        const int owner = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t oLo = (uint32_t)__builtin_amdgcn_readlane((int)segLo, owner);
        const uint32_t oN = (uint32_t)__builtin_amdgcn_readlane((int)nSeg, owner);
        State u;
        u.prevED = k + 1;
        unsigned long long nextKey = lane < oN ? keys[oLo + lane] : 0ull;
        for (uint32_t base = 0; base < oN; base += 64u) {
            const unsigned long long mineKey = nextKey;
            if (base + 64u < oN) nextKey = base + 64u + lane < oN ? keys[oLo + base + 64u + lane] : 0ull; // prefetch
            const uint32_t cnt = min(64u, oN - base);
            uint32_t myRank = FILTER_NONE;
            for (uint32_t j = 0; j < cnt; j++) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mineKey, (int)j);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mineKey >> 32), (int)j);
                uint32_t replaced;
                const uint32_t mine = stepCore(u, base + j, (unsigned long long)lo | ((unsigned long long)hi << 32), replaced);
                if (replaced != FILTER_NONE) {
                    if (replaced >= base) { // still in this chunk's registers
                        if (lane == replaced - base) myRank = FILTER_NONE;
                    } else if (lane == (replaced & 63u)) { // already stored with the previous chunk, by this lane
                        rank[oLo + replaced] = FILTER_NONE;
                    }
                }
                if (lane == j) myRank = mine;
            }
            if (lane < cnt) rank[oLo + base + lane] = myRank;
        }
        if (lane == (uint32_t)owner) mySt.nOut = u.nOut;
    }
    const uint32_t nOut = mySt.nOut;
    if (r < nReads) counts[r] = nOut;
}
__global__ void __launch_bounds__(256)
k_filter_write(const unsigned long long* __restrict__ keys, uint32_t n, const uint64_t* __restrict__ offs, uint32_t k,
               const uint32_t* __restrict__ rank, const uint64_t* __restrict__ outOffs, uint4* __restrict__ out,
               uint32_t* __restrict__ outRead /* read of every occurrence, or null */, uint32_t perStrand, uint32_t layout) {
    const KeyBits kb = keyBits(layout);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t rk = rank[i];
    if (rk == FILTER_NONE) return;
    const unsigned long long key = keys[i];
    const uint32_t grp = (uint32_t)(key >> kb.group);
    const uint32_t r = perStrand ? grp >> 1 : grp;
    const uint32_t len = (uint32_t)(offs[r + 1] - offs[r]);
    const uint32_t begin = (uint32_t)(key >> kb.begin), dist = (uint32_t)(key >> kb.dist) & kb.distMask;
    const uint32_t width = layout == 1u ? len : len - k + ((uint32_t)(key >> 1) & kb.wMask), strand = perStrand ? (grp & 1u) : ((uint32_t)key & 1u);
    out[outOffs[grp] + rk] = make_uint4(begin, begin + width, dist, strand);
    if (outRead) outRead[outOffs[grp] + rk] = r;
}

// ------------------------------------------------------------------ alignments of the final occurrences (§8f rank 1)
// CIGAR of every occurrence that left the filter = IBitParallelED::findCIGAR (bitparallelmatrix.h:460-527): the read
// is aligned with text[begin, end) on a fresh matrix (maxED = the occurrence's distance, first column 0, 1, 2, ...:
// exactly an in-text verification with a fixed start) and traced back from the last cell — horizontal (I) before
// diagonal (M) before vertical (D).  The reference gives in-text occurrences the CIGAR of their verification's
// traceBack (:531-586) instead; that is the same string: along the traced path both matrices hold the same values
// and take the same decisions (tests/test_oracle_golden.py checks it on the reference's own traceback vectors).
// Output per occurrence: n runs of (length << 2 | op), op 0 M / 1 I / 2 D, stored END to BEGIN at ops[occ * stride + j].
// Sequence assignment (IndexInterface::findSeqName, indexinterface.cpp:799-832): seq = the sequence `begin` lies in;
// spans = the occurrence runs over its end (the host trims or drops those, :833-899).
struct AlnRec {
    uint32_t seqId, seqBegin, nOps, spans;
};
constexpr uint32_t CIG_M = 0, CIG_I = 1, CIG_D = 2;
constexpr uint32_t CIGAR_BLOCK_WORDS_MAX_ED = 9; // k_cigar up to this distance, k_cigar_wide beyond (see there)
template <bool NARROW, bool PACKED>
__global__ void __launch_bounds__(256)
k_cigar(DevIndex ix, const uint64_t* __restrict__ offs, MFull mf, const uint4* __restrict__ occs,
        const uint32_t* __restrict__ occRead, uint64_t nOcc, VPlanes V, const uint32_t* __restrict__ seqStarts, uint32_t nSeqs,
        uint16_t* __restrict__ ops, uint32_t stride, AlnRec* __restrict__ aln, uint32_t* __restrict__ flagWord,
        uint32_t gapless /* Hamming distance / exact matches: every CIGAR is <len>M (fmindex.cpp:367, indexinterface.cpp:988) */) {
    __shared__ uint64_t wW[NARROW ? 1 : TBW][256];
    __shared__ uint32_t wN[NARROW ? 16 : 1][256];
    __shared__ uint64_t Ml[ML_WORDS];
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tid = threadIdx.x;
    uint32_t flags = 0, dummyRows = 0;
    const uint64_t strideT = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t HP0 = (~0ull) << MXW_LEFT;
    for (uint64_t base = slot & ~63u; base < nOcc; base += strideT) { // wave-uniform trip count
        const uint64_t it = base + (tid & 63u);
        uint4 o = make_uint4(0, 0, 0, 0);
        uint32_t rs = 0, len = 0, size = 0, col = 0;
        bool have = false;
        if (it < nOcc) {
            o = occs[it];
            rs = 2u * occRead[it] + (o.w & 1u);
            len = (uint32_t)(offs[(rs >> 1) + 1] - offs[rs >> 1]);
            size = o.y - o.x;
            have = true;
            // sequence of the occurrence: last start position <= begin (upper_bound - 1)
            uint32_t lo = 0, hi = nSeqs; // seqStarts[0] == 0; seqStarts[nSeqs] = end of the last sequence
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (seqStarts[mid] <= o.x) lo = mid;
                else hi = mid;
            }
            aln[it].seqId = lo;
            aln[it].seqBegin = o.x - seqStarts[lo];
            aln[it].spans = o.y > seqStarts[lo + 1] ? 1u : 0u;
        }
        const uint32_t maxED = o.z;
        MatGeom g;
        g.n = len + 1;
        g.maxED = maxED;
        g.Wv = maxED;
        g.Wh = maxED;
        g.m = max(g.Wv + g.n, g.Wv + g.Wh + 1u); // (bitparallelmatrix.cpp:98-103: reads shorter than the band)
        col = len;
        bool trace = have && maxED > 0 && size > 0 && !gapless;
        if (trace && NARROW && maxED > TBN_MAX_ED) { // (the host picks the wide kernel for k > 4)
            flags |= FLAG_CAPACITY;
            trace = false;
        }
        uint32_t dummyMask;
        uint64_t ep, eph;
        // rows 1..size of the fresh matrix (all valid: an alignment within maxED exists), trace rows to the lane's slab
        if (trace && maxED > CIGAR_BLOCK_WORDS_MAX_ED) { // (the host launches k_cigar_wide beyond: the match words of 32-row blocks end there)
            flags |= FLAG_CAPACITY;
            trace = false;
        }
        const uint32_t rowsDone = forwardPass<true, NARROW, PACKED>(ix, mf, rs, g, 1u, o.x, trace ? size : 0u, maxED, 0, dummyMask,
                                                                    ep, eph, V, slot, dummyRows, Ml);
        if (trace && rowsDone < size) { // (a row without a cell <= maxED: the occurrence is not an alignment within its distance)
            flags |= FLAG_CAPACITY;
            trace = false;
        }
        uint32_t nOps = 0;
        uint16_t* myOps = ops + it * stride;
        if (have && !trace) { // distance 0 (or an empty range): len x M
            if (len) myOps[nOps++] = (uint16_t)((len << 2) | CIG_M);
        }
        if (trace) {
            uint32_t ti = size, tj = col;
            uint32_t curG = 0xFFFFFFFFu, state = 3u, run = 0;
            auto emit = [&](uint32_t op) {
                if (op != state) {
                    if (run) {
                        if (nOps < stride) myOps[nOps] = (uint16_t)((run << 2) | state);
                        nOps++;
                    }
                    state = op;
                    run = 0;
                }
                run++;
            };
            while (tj > 0) {
                const uint32_t rel = tj + (NARROW ? TB_BELOW : TBW_BELOW) - ti;
                bool hpBit, dgBit;
                if (NARROW) {
                    uint32_t wn = packTraceRowNarrow(0, (~0u) << MX32_LEFT, 0u);
                    if (ti > 0) {
                        const uint32_t gq = (ti - 1) >> 4, jq = (ti - 1) & 15u;
                        if (gq != curG) {
                            curG = gq;
                            const uint4* L = traceLine(V, slot, gq);
#pragma unroll
                            for (int h = 0; h < 4; h++) {
                                const uint4 v = L[h];
                                wN[4 * h][tid] = v.x;
                                wN[4 * h + 1][tid] = v.y;
                                wN[4 * h + 2][tid] = v.z;
                                wN[4 * h + 3][tid] = v.w;
                            }
                        }
                        wn = wN[jq][tid];
                    }
                    if (rel < TBN_REL_LO || rel > TBN_REL_HI) {
                        flags |= FLAG_CAPACITY;
                        break;
                    }
                    hpBit = rel >= TBN_HP_LO && ((wn >> (rel - TBN_HP_LO)) & 1u);
                    dgBit = rel == TBN_REL_HI || ((wn >> (16u + rel - TBN_DG_LO)) & 1u);
                } else {
                    uint64_t ww = packTraceRow(0, HP0, 0ull);
                    if (ti > 0) {
                        const uint32_t gq = (ti - 1) >> 3, jq = (ti - 1) & 7u;
                        if (gq != curG) {
                            curG = gq;
                            const uint4* L = traceLine(V, slot, gq);
#pragma unroll
                            for (int h = 0; h < 4; h++) {
                                const uint4 v = L[h];
                                wW[2 * h][tid] = (uint64_t)v.x | ((uint64_t)v.y << 32);
                                wW[2 * h + 1][tid] = (uint64_t)v.z | ((uint64_t)v.w << 32);
                            }
                        }
                        ww = wW[jq][tid];
                    }
                    if (rel > 31u) {
                        flags |= FLAG_CAPACITY;
                        break;
                    }
                    hpBit = ((uint32_t)ww >> rel) & 1u;
                    dgBit = ((uint32_t)(ww >> 32) >> rel) & 1u;
                }
                if (hpBit) { // gap in horizontal direction: insertion (:488-494)
                    --tj;
                    emit(CIG_I);
                } else if (ti > 0 && dgBit) { // diagonal (:496-506)
                    --tj;
                    --ti;
                    emit(CIG_M);
                } else { // gap in vertical direction (:508-514)
                    if (ti == 0) { // (cannot happen: row 0 only has horizontal steps)
                        flags |= FLAG_CAPACITY;
                        break;
                    }
                    --ti;
                    emit(CIG_D);
                }
            }
            for (; ti > 0; --ti) emit(CIG_D); // findCIGAR walks on to (0, 0): leading deletions
            if (run) {
                if (nOps < stride) myOps[nOps] = (uint16_t)((run << 2) | state);
                nOps++;
            }
            if (nOps > stride) flags |= FLAG_CAPACITY;
        }
        if (have) aln[it].nOps = nOps;
    }
    if (flags) atomicOr(flagWord, flags);
}

// The same for batches beyond 9 errors.  k_cigar takes its match words from the 64-bit words of 32-row blocks (k_match_words), which reach
// 40 columns past the block's first row: enough for a band of Wh <= 9 columns right of the diagonal in the block's last row, not for the
// Wh = distance <= 13 of these alignments.  Here the rows come from the matrix with the wide left margin and match words taken straight from
// the read's bit-strings (WideRow<WxTen>: a fixed start has Wv = Wh = distance <= 13), stored as in k_verify_wide — 16 bytes {HP, M | ~D0} per
// row, the rows of a wavefront contiguous — and findCIGAR's walk (bitparallelmatrix.h:480-522) reads them from the last cell to (0, 0).
__global__ void __launch_bounds__(256)
k_cigar_wide(DevIndex ix, const uint64_t* __restrict__ offs, const uint32_t* __restrict__ G, uint32_t gw, const uint4* __restrict__ occs,
             const uint32_t* __restrict__ occRead, uint64_t nOcc, uint8_t* __restrict__ slab, uint32_t slotBytes,
             const uint32_t* __restrict__ seqStarts, uint32_t nSeqs, uint16_t* __restrict__ ops, uint32_t stride, AlnRec* __restrict__ aln,
             uint32_t* __restrict__ flagWord, uint32_t gapless) {
    typedef WxTen WX;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint4* const rowBits = reinterpret_cast<uint4*>(slab + (size_t)(slot - lane) * slotBytes) + lane;
    uint32_t flags = 0;
    const uint64_t strideT = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = slot - lane; base < nOcc; base += strideT) { // wave-uniform trip count
        const uint64_t it = base + lane;
        if (it >= nOcc) continue;
        const uint4 o = occs[it];
        const uint32_t rs = 2u * occRead[it] + (o.w & 1u);
        const uint32_t len = (uint32_t)(offs[(rs >> 1) + 1] - offs[rs >> 1]);
        const uint32_t size = o.y - o.x, maxED = o.z, col = len;
        { // sequence of the occurrence: last start position <= begin (upper_bound - 1)
            uint32_t lo = 0, hi = nSeqs; // seqStarts[0] == 0; seqStarts[nSeqs] = end of the last sequence
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (seqStarts[mid] <= o.x) lo = mid;
                else hi = mid;
            }
            aln[it].seqId = lo;
            aln[it].seqBegin = o.x - seqStarts[lo];
            aln[it].spans = o.y > seqStarts[lo + 1] ? 1u : 0u;
        }
        uint32_t nOps = 0;
        uint16_t* myOps = ops + it * stride;
        bool trace = maxED > 0 && size > 0 && !gapless;
        const MatGeom g = wideGeom(len, maxED, 1u);
        if (trace && (g.Wv >= WX::LEFT || WX::DIAG + g.Wh + WX::BLOCK > 63u || (size + 1u) * VW_ROW_BYTES > slotBytes)) {
            flags |= FLAG_CAPACITY;
            trace = false;
        }
        if (!trace) { // distance 0 (or an empty range): len x M
            if (len) myOps[nOps++] = (uint16_t)((len << 2) | CIG_M);
            aln[it].nOps = nOps;
            continue;
        }
        WideRow<WX> mx;
        mx.init(g, 1u);
        rowBits[0] = make_uint4((uint32_t)mx.HP, (uint32_t)(mx.HP >> 32), ~0u, ~0u);
        bool ok = true;
        for (uint32_t r = 1; r <= size; r++) { // rows 1 .. size of the fresh matrix (all valid: an alignment within maxED exists)
            if (r == 1u || (r & 15u) == 0u) mx.loadBlock(ix, G, gw, rs, len, o.x, r >> 4);
            uint64_t M, D0;
            ok = mx.step(g, r, M, D0) && ok;
            const uint64_t md = M | ~D0;
            rowBits[(size_t)r * 64u] = make_uint4((uint32_t)mx.HP, (uint32_t)(mx.HP >> 32), (uint32_t)md, (uint32_t)(md >> 32));
        }
        if (!ok) flags |= FLAG_CAPACITY; // (a row without a cell <= maxED: the occurrence is not an alignment within its distance)
        uint32_t ti = size, tj = col, state = 3u, run = 0;
        auto emit = [&](uint32_t op) {
            if (op != state) {
                if (run) {
                    if (nOps < stride) myOps[nOps] = (uint16_t)((run << 2) | state);
                    nOps++;
                }
                state = op;
                run = 0;
            }
            run++;
        };
        while (ok && tj > 0) {
            const uint32_t bitIdx = (tj - (ti / WX::BLOCK) * WX::BLOCK) + WX::DIAG; // (unsigned arithmetic as in the reference, :486)
            if (bitIdx >= 64u) {
                flags |= FLAG_CAPACITY;
                break;
            }
            const uint4 rb = rowBits[(size_t)ti * 64u];
            const uint64_t hp = rb.x | ((uint64_t)rb.y << 32), md = rb.z | ((uint64_t)rb.w << 32);
            if ((hp >> bitIdx) & 1ull) { // gap in horizontal direction: insertion (:488-494)
                --tj;
                emit(CIG_I);
            } else if (ti > 0 && ((md >> bitIdx) & 1ull)) { // diagonal (:496-506)
                --tj;
                --ti;
                emit(CIG_M);
            } else { // gap in vertical direction (:508-514)
                if (ti == 0) { // (row 0 only has horizontal steps)
                    flags |= FLAG_CAPACITY;
                    break;
                }
                --ti;
                emit(CIG_D);
            }
        }
        for (; ok && ti > 0; --ti) emit(CIG_D); // findCIGAR walks on to (0, 0): leading deletions
        if (run) {
            if (nOps < stride) myOps[nOps] = (uint16_t)((run << 2) | state);
            nOps++;
        }
        if (nOps > stride) flags |= FLAG_CAPACITY;
        aln[it].nOps = nOps;
    }
    if (flags) atomicOr(flagWord, flags);
}

} // namespace cmb

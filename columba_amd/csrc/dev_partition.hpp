// Device-side PROLOGUE of the search: everything matchWithSearches (reference src/searchstrategy.cpp:425-493)
// does before the approximate DFS is a chain of single-character bidirectional extensions.  It runs as two
// kernels whose loop bodies are small:
//   k_parts  one lane per read x strand, PartMachine below: partitioning — uniform / static
//            (calculateExactMatchRanges :158-190) or dynamic (seed :381-419 + greedy extension :299-379) —
//            and the dynamic scheme selection (src/searchstrategy.h:2505-2537).  Output: the parts, the
//            exact-match range pair of every part, the selected scheme.
//   k_exact  one lane per (read x strand, search of the selected scheme) + one per read x strand for the
//            part-level in-text pre-verification (:464-476): the exact phases of doRecSearch (:1181-1254) and
//            the entry decision of recApproxMatchEditEntry (src/indexinterface.cpp:1306-1325); for k = 0
//            one lane per read x strand runs exactMatchesOutput (src/indexinterface.cpp:947-1014).
// Both follow the same rule: every lane of a wavefront is at the SAME program point when it touches memory
// (one "memory step" per loop iteration: rank blocks of an extension, k-mer table entries, or a record),
// whatever logical phase its read is in.  What needs the irregular DFS is emitted as a compact DfsTask for
// k_dfs; what can be verified in the text is emitted as work items for k_verify.
#pragma once
#include "dev_search.hpp"

namespace cmb {

struct DfsTask {
    uint32_t rsId;
    uint8_t scheme, search, idx, pad; // selected scheme, search number, first approximate phase
    RangePair r;                      // start range (after the exact phases)
    uint32_t depth;                   // exact length matched so far
};

template <int MP> struct PartOutT { // per read x strand: the parts, needed again by k_dfs
    uint16_t pb[MP], pe[MP];
};
typedef PartOutT<MAXP> PartOut;

enum : int { PH_CALC = 0, PH_DYN, PH_SEED, PH_FIN, PH_DONE };
// what a lane asks of the iteration's single memory step
enum : int { RQ_NONE = 0, RQ_RANK, RQ_SEED, RQ_REC };

// PARTITION: 0 uniform, 1 static, 2 dynamic (searchstrategy.h PartitionStrategy) — a template parameter, so that
// each k_parts instance only carries the phases of its own mode (the extension loop of the dynamic mode is a
// third of the instructions of the generic one).
template <int PARTITION, int MP = MAXP>
struct PartMachine {
    const DevIndex& ix;
    const DevStrategyKT<MP>& st;
    // task
    uint32_t rsId = 0, len = 0, k = 0;
    const uint8_t* seq = nullptr;
    // per-lane partition state lives in LDS (dynamic indexing would otherwise force it into
    // scratch memory, i.e. a memory round trip per access): layout [field][part][lane]
    uint32_t* lds;   // base of this block's LDS slab
    uint32_t ltid;   // lane index inside the block
    uint32_t lstride; // = blockDim.x
    uint32_t lparts;  // parts per field (numParts of the strategy)
    __device__ __forceinline__ uint32_t& L(int field, int part) const {
        return lds[((uint32_t)field * lparts + (uint32_t)part) * lstride + ltid];
    }
    __device__ __forceinline__ uint32_t PB(int i) const { return L(4, i) & 0xFFFFu; }
    __device__ __forceinline__ uint32_t PE(int i) const { return L(4, i) >> 16; }
    __device__ __forceinline__ void setPB(int i, uint32_t v) { L(4, i) = (L(4, i) & 0xFFFF0000u) | (v & 0xFFFFu); }
    __device__ __forceinline__ void setPE(int i, uint32_t v) { L(4, i) = (L(4, i) & 0xFFFFu) | (v << 16); }
    __device__ __forceinline__ void setPBE(int i, uint32_t b, uint32_t e) { L(4, i) = (b & 0xFFFFu) | (e << 16); }
    __device__ __forceinline__ RangePair EX(int i) const {
        return RangePair{{L(0, i), L(1, i)}, {L(2, i), L(3, i)}};
    }
    __device__ __forceinline__ void setEX(int i, const RangePair& r) {
        L(0, i) = r.sa.b;
        L(1, i) = r.sa.e;
        L(2, i) = r.rev.b;
        L(3, i) = r.rev.e;
    }
    __device__ __forceinline__ uint32_t EXW(int i) const {
        const uint32_t b = L(0, i), e = L(1, i);
        return e <= b ? 0u : e - b;
    }
    // counters
    uint32_t cNode = 0, cExp = 0, flags = 0;
    // machine state
    int phase = PH_DONE;
    int numParts = 0;
    int pi = 0;          // PH_CALC: current part
    uint32_t ci = 0;     // PH_CALC: next character index inside the part
    uint32_t cend = 0;   // PH_CALC: number of characters to match in the part
    uint32_t cb = 0, ce = 0; // PH_CALC: sub-range of the read being matched
    int cdir = 0;        // direction of the substring being matched
    uint32_t j = 0;      // PH_DYN: assigned characters so far
    int partToExtend = 0, dynDir = 0;
    // pending request for the memory step (RQ_RANK: extension of reqParent with reqCode)
    int req = RQ_NONE;
    int reqMode = 0;
    uint32_t reqCode = 0;
    RangePair reqParent;

    __device__ PartMachine(const DevIndex& i, const DevStrategyKT<MP>& s, uint32_t* ldsBase, uint32_t tid, uint32_t stride)
        : ix(i), st(s), lds(ldsBase), ltid(tid), lstride(stride), lparts(s.numParts ? s.numParts : 1) {}
    __device__ void setReadWords(uint32_t maxLen) { // LDS words after the 5 x lparts partition fields
        rdBase = 5u * lparts;
        pw1 = (maxLen + 31) / 32;
    }

    // ---- helpers ----------------------------------------------------------------------------
    // The read is kept in LDS as two bit-strings (low / high bit of code-1), built at begin() from the
    // match bit-strings k_prep wrote, so that the character of an extension costs no global load.
    // Reads with a non-ACGT character (hasN) take the byte path.
    uint32_t rdBase = 0, pw1 = 1;
    bool hasN = false;
    __device__ __forceinline__ uint32_t& RD(uint32_t plane, uint32_t w) const {
        return lds[(rdBase + plane * pw1 + w) * lstride + ltid];
    }
    __device__ __forceinline__ uint32_t code(uint32_t i) const {
        if (hasN) return seq[i];
        const uint32_t w = i >> 5, b = i & 31u;
        return 1u + ((RD(0, w) >> b) & 1u) + 2u * ((RD(1, w) >> b) & 1u);
    }
    __device__ __forceinline__ uint32_t charAt(uint32_t b, uint32_t e, int d, uint32_t i) const {
        return code(d == 0 ? b + i : e - i - 1);
    }
    __device__ __forceinline__ uint32_t kmerKey(uint32_t begin, uint32_t end, bool& valid) const {
        valid = true;
        uint32_t key = 0;
        if (hasN) {
            for (uint32_t i = begin; i < end; i++)
                if (seq[i] > 4) valid = false;
            if (!valid) return 0;
            for (uint32_t i = 0; i < ix.kmerSize; i++) key = (key << 2) | (uint32_t)(seq[begin + i] - 1);
            return key;
        }
        const uint32_t w = begin >> 5, b = begin & 31u;
        const bool more = w + 1 < pw1;
        const uint64_t lo = ((uint64_t)RD(0, w) | ((uint64_t)(more ? RD(0, w + 1) : 0u) << 32)) >> b;
        const uint64_t hi = ((uint64_t)RD(1, w) | ((uint64_t)(more ? RD(1, w + 1) : 0u) << 32)) >> b;
        for (uint32_t i = 0; i < ix.kmerSize; i++)
            key = (key << 2) | (uint32_t)(((hi >> i) & 1ull) << 1) | (uint32_t)((lo >> i) & 1ull);
        return key;
    }
    __device__ __forceinline__ RangePair kmer(uint32_t begin, uint32_t end) const { // indexinterface.h:590
        bool valid;
        const uint32_t key = kmerKey(begin, end, valid);
        if (!valid) return RangePair{{0, 0}, {0, 0}};
        const uint4 v = ix.kmer[key];
        return RangePair{{v.x, v.y}, {v.z, v.w}};
    }
    __device__ __forceinline__ void request(int mode, const RangePair& parent, uint32_t code) {
        req = RQ_RANK;
        reqMode = mode;
        reqParent = parent;
        reqCode = code;
    }

    // ---- task start -------------------------------------------------------------------------
    // The read record k_prep wrote (len | hasN << 16, then (low, high) code-bit word pairs) has arrived
    // in v[]: unpack it into LDS and start the prologue of read x strand `rs`.
    template <bool LONG> // LONG: reads beyond 256 characters (a kernel instance of its own: the common one carries no code for them)
    __device__ void begin(uint32_t rs, const uint4 v[5], const uint4* recBase, uint32_t recQ, const uint8_t* s, uint32_t kk) {
        rsId = rs;
        seq = s;
        k = kk;
        {
            const uint32_t* vw = reinterpret_cast<const uint32_t*>(v);
            len = vw[0] & 0xFFFFu;
            hasN = (vw[0] >> 16) & 1u;
#pragma unroll
            for (uint32_t w = 0; w < 8; w++) // the words of the first 256 characters arrive in registers
                if (w < pw1) { // record words: header, then (low, high) pairs
                    RD(0, w) = vw[1 + 2 * w];
                    RD(1, w) = vw[2 + 2 * w];
                }
            if (LONG && pw1 > 8) { // longer reads (up to MAX_READ): the rest straight from the record
                const uint32_t* recWords = reinterpret_cast<const uint32_t*>(recBase + (size_t)rs * recQ);
                for (uint32_t w = 8; w < pw1; w++) {
                    RD(0, w) = recWords[1 + 2 * w];
                    RD(1, w) = recWords[2 + 2 * w];
                }
            }
        }
        req = RQ_NONE;
        partToExtend = 0;
        dynDir = 0;
        numParts = st.numParts;
        if (numParts >= (int)len || numParts == 1 || len > (uint32_t)MAX_READ) {
            flags |= FLAG_UNSUPPORTED_READ; // naive fallback (searchstrategy.cpp:148-152) not on device
            phase = PH_DONE;
            return;
        }
        const uint32_t L = len;
        if (PARTITION == 0) { // partitionUniform (:194-209)
            for (int i = 0; i < numParts; i++) {
                const uint32_t b = (uint32_t)((i * 1.0 / numParts) * L);
                uint32_t e = (uint32_t)(((i + 1) * 1.0 / numParts) * L);
                setPBE(i, b, e > L ? L : e);
            }
            setPE(numParts - 1, L);
            startCalcPart(0);
            phase = PH_CALC;
        } else if (PARTITION == 1) { // setParts (:221-238)
            const int pSize = (int)L;
            const double* bg = st.begins;
            setPBE(0, 0, (uint32_t)(bg[0] * pSize) & 0xFFFFu);
            for (int i = 0; i < numParts - 2; i++)
                setPBE(i + 1, (uint32_t)(bg[i] * pSize), (uint32_t)(bg[i + 1] * pSize) & 0xFFFFu);
            setPBE(numParts - 1, (uint32_t)(bg[numParts - 2] * pSize), L);
            for (int i = 0; i < numParts; i++)
                if (PE(i) > L) setPE(i, L); // Substring::check()
            startCalcPart(0);
            phase = PH_CALC;
        } else { // seed (:381-419)
            const uint32_t ws = ix.kmerSize;
            const bool useKmer = ((uint32_t)numParts * ws < (L * 2) / 3) && (L >= st.kmerCutOff);
            const int wSize = useKmer ? (int)ws : 1;
            setPBE(0, 0, (uint32_t)wSize);
            for (int i = 1; i < numParts - 1; i++) {
                const uint32_t b = (uint32_t)(uint16_t)(int)((st.seeding[i - 1] * L) - (wSize / 2));
                setPBE(i, b, (b + wSize) & 0xFFFFu);
            }
            setPBE(numParts - 1, L - wSize, L);
            for (int i = 0; i + 1 < numParts; i++)
                if (PE(i) > PB(i + 1)) { // `assert(parts[i].end() <= parts[i + 1].begin())` (:404-407)
                    flags |= FLAG_SEED_OVERLAP;
                    phase = PH_DONE;
                    return;
                }
            j = (uint32_t)(numParts * wSize);
            if (useKmer) { // the k-mer table entries of all parts are fetched together in the memory step
                phase = PH_SEED;
                req = RQ_SEED;
                return;
            }
            for (int i = 0; i < numParts; i++) { // getRangeOfSingleChar (fmindex.cpp:434-445)
                const uint32_t code = this->code(PB(i));
                if (code < 1 || code > 4) setEX(i, RangePair{{0, 0}, {0, 0}});
                else {
                    const uint32_t lo = ix.counts[code], hi = code < 4 ? ix.counts[code + 1] : ix.n;
                    setEX(i, RangePair{{lo, hi}, {lo, hi}});
                }
            }
            phase = PH_DYN;
        }
    }

    // RQ_SEED: issue the k-mer table loads of all parts (indexinterface.h:590), then take the replies
    __device__ __forceinline__ void seedIssue(uint4 v[MP]) const {
#pragma unroll
        for (int i = 0; i < MP; i++)
            if (i < numParts) {
                bool valid;
                const uint32_t key = kmerKey(PB(i), PE(i), valid);
                v[i] = make_uint4(0, 0, 0, 0);
                if (valid) v[i] = ix.kmer[key];
            }
    }
    __device__ __forceinline__ void seedTake(const uint4 v[MP]) {
#pragma unroll
        for (int i = 0; i < MP; i++)
            if (i < numParts) setEX(i, RangePair{{v[i].x, v[i].y}, {v[i].z, v[i].w}});
        phase = PH_DYN;
    }

    // calculateExactMatchRanges (:158-190): stages 0..P-1 match EVERY part forwards (the loop at :166
    // runs over all parts), stage P re-matches the last part backwards, uni-directionally (:178-189)
    __device__ void startCalcPart(int stage) {
        pi = stage;
        const int i = stage < numParts ? stage : numParts - 1;
        const uint32_t ws = ix.kmerSize;
        const uint32_t b = PB(i), e = PE(i);
        const uint32_t size = e > b ? e - b : 0;
        const bool last = (stage == numParts);
        if (!last) {
            const uint32_t start = b + (size >= ws ? ws : 0);
            setEX(i, size >= ws ? kmer(b, start) : RangePair{{0, ix.n}, {0, ix.n}});
            cb = start;
            ce = e;
            cdir = 0;
        } else {
            const uint32_t end = size >= ws ? e - ws : e;
            setEX(i, size >= ws ? kmer(end, e) : RangePair{{0, ix.n}, {0, ix.n}});
            cb = b;
            ce = end;
            cdir = 1;
        }
        ci = 0;
        cend = ce > cb ? ce - cb : 0;
    }

    // ---- one scheduling step: runs until an extension is requested or the task is finished ----
    __device__ void advance() {
        for (;;) {
            if (PARTITION != 2 && phase == PH_CALC) {
                const int part = pi < numParts ? pi : numParts - 1;
                if (ci < cend) {
                    const uint32_t code = charAt(cb, ce, cdir, ci);
                    if (code >= 1 && code <= 4) {
                        request(pi == numParts ? 2 : 0, EX(part), code);
                        return;
                    }
                    setEX(part, RangePair{{0, 0}, {0, 0}}); // addChar on a non-ACGT character
                }
                // part finished (or failed): next stage
                if (pi < numParts) {
                    startCalcPart(pi + 1);
                } else {
                    phase = PH_FIN;
                }
                continue;
            }
            if (PARTITION == 2 && phase == PH_DYN) { // partitionDynamic loop body (:324-378)
                if (j >= len) {
                    phase = PH_FIN;
                    continue;
                }
                uint64_t maxRangeWeighted = 0;
                {
                    uint32_t prevPE = 0, prevW = 0;
                    uint32_t curPBE = L(4, 0), curW = EXW(0);
                    for (int i = 0; i < numParts; i++) {
                        const bool lastPart = (i == numParts - 1);
                        const uint32_t nextPBE = lastPart ? 0u : L(4, i + 1);
                        const uint32_t nextW = lastPart ? 0u : EXW(i + 1);
                        const uint32_t b = curPBE & 0xFFFFu, e = curPBE >> 16;
                        const bool noLeft = (i == 0) || b == prevPE;
                        const bool noRight = lastPart || e == (nextPBE & 0xFFFFu);
                        if (!(noLeft && noRight)) {
                            const uint64_t wv = (uint64_t)curW * st.weights[i];
                            if (wv > maxRangeWeighted) {
                                maxRangeWeighted = wv;
                                partToExtend = i;
                                if (noLeft) dynDir = 0;
                                else if (noRight) dynDir = 1;
                                else dynDir = (prevW < nextW) ? 1 : 0;
                            }
                        }
                        prevPE = e;
                        prevW = curW;
                        curPBE = nextPBE;
                        curW = nextW;
                    }
                }
                if (maxRangeWeighted == 0) { // extendParts (:283-297)
                    for (int i = 0; i < numParts; i++) {
                        if (i != numParts - 1 && PE(i) != PB(i + 1)) setPE(i, PB(i + 1));
                        if (i != 0 && PB(i) != PE(i - 1)) setPB(i, PE(i - 1));
                    }
                    phase = PH_FIN;
                    continue;
                }
                uint32_t code;
                if (dynDir == 0) {
                    const uint32_t e = PE(partToExtend) + 1;
                    setPE(partToExtend, e);
                    code = this->code(e - 1);
                } else {
                    const uint32_t b = PB(partToExtend) - 1;
                    setPB(partToExtend, b);
                    code = this->code(b);
                }
                j++;
                if (code >= 1 && code <= 4) {
                    request(partToExtend == numParts - 1 ? 2 : dynDir, EX(partToExtend), code);
                    return;
                }
                setEX(partToExtend, RangePair{{0, 0}, {0, 0}});
                continue;
            }
            return; // PH_FIN (outputs are written by the kernel), PH_SEED, PH_DONE
        }
    }

    // ---- consume the result of the requested extension ----------------------------------------
    __device__ void resume(bool ok, const RangePair& child) {
        cExp++;
        if (ok) cNode++;
        const RangePair res = ok ? child : RangePair{{0, 0}, {0, 0}};
        switch (phase) {
        case PH_CALC:
            setEX(pi < numParts ? pi : numParts - 1, res);
            if (ok) ci++;
            else ci = cend; // matchStringBidirectionally stops at the first failure
            break;
        case PH_DYN:
            setEX(partToExtend, res);
            break;
        default:
            break;
        }
    }

    // partitioning done: select the scheme (MultipleSchemes::createSearches, searchstrategy.h:2505-2537) and
    // hand parts, exact ranges and selection to k_exact
    __device__ void finish(PartOutT<MP>* parts, uint4* exr, uint8_t* psel, uint32_t total) {
        int sel = 0;
        if (st.nSchemes > 1) {
            uint32_t tot = 0;
            for (int i = 0; i < numParts; i++) tot += EXW(i);
            if (tot > (uint32_t)numParts) {
                uint32_t minValue = EXW(st.sch[0].critical);
                for (int i = 1; i < st.nSchemes; i++) {
                    const uint32_t w = EXW(st.sch[i].critical);
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
            po.pb[i] = i < numParts ? (uint16_t)PB(i) : (uint16_t)0;
            po.pe[i] = i < numParts ? (uint16_t)PE(i) : (uint16_t)0;
        }
        parts[rsId] = po;
        for (int i = 0; i < numParts; i++) {
            const RangePair r = EX(i);
            exr[(size_t)i * total + rsId] = make_uint4(r.sa.b, r.sa.e, r.rev.b, r.rev.e);
        }
        psel[rsId] = (uint8_t)sel;
        phase = PH_DONE;
    }
};

// ---- k_exact: one lane per (read x strand, slot) -----------------------------------------------------
// slot < nSlots - 1 : search `slot` of the selected scheme (exact phases, then items or a DFS task);
// slot = nSlots - 1 : the part-level in-text pre-verification (searchstrategy.cpp:464-476);
// k = 0            : one lane per read x strand, exactMatchesOutput (indexinterface.cpp:947-1014).
enum : int { EX_IDLE = 0, EX_HDR, EX_LOAD, EX_RUN, EX_K0 };

template <int MP = MAXP>
struct ExactLane {
    const DevIndex& ix;
    const DevStrategyKT<MP>& st;
    // LDS per lane: lparts words (pb | pe << 16), then 2 x pw1 read words (low / high code bits)
    uint32_t* lds;
    uint32_t ltid, lstride, lparts, pw1;
    __device__ __forceinline__ uint32_t& PBE(int i) const { return lds[(uint32_t)i * lstride + ltid]; }
    __device__ __forceinline__ uint32_t PB(int i) const { return PBE(i) & 0xFFFFu; }
    __device__ __forceinline__ uint32_t PE(int i) const { return PBE(i) >> 16; }
    __device__ __forceinline__ uint32_t& RD(uint32_t plane, uint32_t w) const {
        return lds[(lparts + plane * pw1 + w) * lstride + ltid];
    }
    // task
    uint32_t rsId = 0, slot = 0, len = 0;
    const uint8_t* seq = nullptr;
    bool hasN = false;
    int phase = EX_IDLE;
    int sel = 0, partInSearch = 0;
    uint32_t ci = 0, exactLength = 0, k0i = 0;
    RangePair cur;
    // pending extension
    bool req = false;
    int reqMode = 0;
    uint32_t reqCode = 0;
    // staged output (at most one group / task per iteration)
    uint32_t stN = 0, stB = 0, stA = 0, stMeta = 0;
    bool stDfs = false;
    uint32_t stIdx = 0, stDepth = 0;
    RangePair stR;
    // counters
    uint32_t cNode = 0, cExp = 0, cImm = 0, cStart = 0;

    __device__ ExactLane(const DevIndex& i, const DevStrategyKT<MP>& s, uint32_t* ldsBase, uint32_t tid, uint32_t stride,
                         uint32_t maxLen)
        : ix(i), st(s), lds(ldsBase), ltid(tid), lstride(stride), lparts(s.numParts ? s.numParts : 1),
          pw1((maxLen + 31) / 32) {}

    __device__ __forceinline__ uint32_t code(uint32_t i) const {
        if (hasN) return seq[i];
        const uint32_t w = i >> 5, b = i & 31u;
        return 1u + ((RD(0, w) >> b) & 1u) + 2u * ((RD(1, w) >> b) & 1u);
    }
    __device__ __forceinline__ void emitItems(const Range& sa, uint32_t a, uint32_t meta) {
        stN = sa.width();
        stB = sa.b;
        stA = a;
        stMeta = meta;
    }
    __device__ __forceinline__ void emitDfs(int idx, const RangePair& r, uint32_t depth) {
        stDfs = true;
        stIdx = (uint32_t)idx;
        stR = r;
        stDepth = depth;
    }
    // read record (k_prep): v[0].x = len | hasN << 16, then (low, high) word pairs
    template <bool LONG>
    __device__ __forceinline__ void takeRecord(const uint4 v[5], const uint4* recBase, uint32_t recQ) {
        const uint32_t* vw = reinterpret_cast<const uint32_t*>(v);
        len = vw[0] & 0xFFFFu;
        hasN = (vw[0] >> 16) & 1u;
#pragma unroll
        for (uint32_t w = 0; w < 8; w++)
            if (w < pw1) {
                RD(0, w) = vw[1 + 2 * w];
                RD(1, w) = vw[2 + 2 * w];
            }
        if (LONG && pw1 > 8) { // reads beyond 256 characters: the rest straight from the record
            const uint32_t* recWords = reinterpret_cast<const uint32_t*>(recBase + (size_t)rsId * recQ);
            for (uint32_t w = 8; w < pw1; w++) {
                RD(0, w) = recWords[1 + 2 * w];
                RD(1, w) = recWords[2 + 2 * w];
            }
        }
    }

    // the search starts from the exact range of its first part (doRecSearch, searchstrategy.cpp:1181-1254)
    __device__ void startSearch(const RangePair& first) {
        const DevSearchT<MP>& s = st.sch[sel].s[slot];
        cur = first;
        if (cur.width() <= ix.switchPoint) { // covered by the part-level pre-verification
            phase = EX_IDLE;
            return;
        }
        const int f = s.order[0];
        partInSearch = 1;
        exactLength = PE(f) - PB(f);
        ci = 0;
        phase = EX_RUN;
    }

    // bookkeeping up to the next extension of the running search
    __device__ void advance() {
        const DevSearchT<MP>& s = st.sch[sel].s[slot];
        for (;;) {
            if (s.U[partInSearch] == 0) {
                const int part = s.order[partInSearch];
                const uint32_t b = PB(part), e = PE(part);
                const uint32_t n = e > b ? e - b : 0;
                if (ci < n) {
                    const uint32_t c = code(s.dir[partInSearch] == 0 ? b + ci : e - ci - 1);
                    if (c >= 1 && c <= 4) {
                        const bool uni = s.uniAll || partInSearch >= (int)s.uniIdx;
                        req = true;
                        reqMode = uni ? 2 : (s.dir[partInSearch] == 0 ? 0 : 1);
                        reqCode = c;
                        return;
                    }
                    cur = RangePair{{0, 0}, {0, 0}};
                }
                if (cur.empty()) { // `if (startRange.empty()) return;`
                    phase = EX_IDLE;
                    return;
                }
                exactLength += n;
                partInSearch++;
                ci = 0;
                continue;
            }
            // exact phases done: start approximate matching
            if (st.metric == 1 && cur.width() <= ix.switchPoint) { // recApproxMatchEditEntry
                cImm++;
                const uint32_t bg = PB(s.low[partInSearch - 1]);
                const uint32_t maxEDs = s.U[s.n - 1], minEDs = s.L[s.n - 1];
                emitItems(cur.sa, bg == 0 ? 0 : bg + maxEDs, packMeta(0, maxEDs, minEDs, bg == 0, ITEM_EDIT));
            } else {
                if (st.metric == 1) cStart++;
                emitDfs(partInSearch, cur, exactLength);
            }
            phase = EX_IDLE;
            return;
        }
    }
    __device__ void resume(bool ok, const RangePair& child) {
        cExp++;
        if (ok) cNode++;
        if (phase == EX_RUN) {
            cur = ok ? child : RangePair{{0, 0}, {0, 0}};
            if (ok) ci++;
            else phase = EX_IDLE; // range is empty: this search is over
        } else { // EX_K0
            if (!ok) { // no exact match possible
                phase = EX_IDLE;
                return;
            }
            cur.sa = child.sa;
            k0i--;
            if (cur.sa.width() <= ix.switchPoint) { // switch to in-text verification with k0i chars left
                emitItems(cur.sa, k0i, packMeta(0, 0, 0, 0, ITEM_EXACT));
                phase = EX_IDLE;
            }
        }
    }
    // k = 0: next step of exactMatchesOutput
    __device__ void advanceK0() {
        if (k0i == 0) { // everything matched in the index
            emitItems(cur.sa, 0, packMeta(0, 0, 0, 1, ITEM_EXACT));
            phase = EX_IDLE;
            return;
        }
        const uint32_t c = code(k0i - 1);
        if (c > 4) {
            phase = EX_IDLE;
            return;
        }
        req = true;
        reqMode = 2;
        reqCode = c;
    }
};

} // namespace cmb

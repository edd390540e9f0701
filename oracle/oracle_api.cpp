// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_core.hpp header).
// extern "C" surface of the CPU oracle, loaded with ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never linked into
// or called from the product library.
// ============================================================================
#include "oracle_move.hpp"
#include "oracle_search.hpp"
#include "oracle_move_search.hpp"
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

using namespace orc;

extern "C" {

struct orc_index_desc {
    uint64_t text_length; // n including '$'
    const uint8_t* text;
    uint64_t counts[5];
    uint64_t dollar_pos_fwd;
    const uint64_t* bv_fwd;
    const uint64_t* cnt_fwd;
    uint64_t dollar_pos_rev;
    const uint64_t* bv_rev;
    const uint64_t* cnt_rev;
    const uint64_t* bwt_words; // EncodedText<5> words
    const uint64_t* sa_bv;
    const uint64_t* sa_bv_counts;
    const uint32_t* sa_samples;
    uint32_t sa_sparseness;
    uint32_t switch_point;
    uint32_t kmer_size;
    uint32_t n_seqs;
    const uint32_t* seq_starts;
};

struct orc_occ {
    uint32_t begin, end, distance, strand;
};

struct OrcIndex {
    Index idx;
};

// populateTable indexinterface.cpp:294-335 (forward direction, bidirectional)
static void populateTable(Index& idx) {
    const len_t w = idx.wordSize;
    idx.kmerTable.assign(1ull << (2 * w), RangePair());
    if (w == 0) {
        idx.kmerTable[0] = idx.completeRange();
        return;
    }
    struct Node {
        RangePair r;
        len_t row;
        uint64_t key;
    };
    std::vector<Node> stack;
    auto extend = [&](const RangePair& parent, len_t row, uint64_t key) {
        for (len_t i = 1; i < 5; ++i) {
            RangePair child;
            if (idx.extendForward(i, parent, child))
                stack.push_back({child, row + 1, (key << 2) | (i - 1)});
        }
    };
    extend(idx.completeRange(), 0, 0);
    while (!stack.empty()) {
        Node cur = stack.back();
        stack.pop_back();
        if (cur.row == w)
            idx.kmerTable[cur.key] = cur.r;
        else
            extend(cur.r, cur.row, cur.key);
    }
}

void* orc_index_create(const orc_index_desc* d) {
    OrcIndex* h = new OrcIndex();
    Index& x = h->idx;
    x.textLength = (len_t)d->text_length;
    x.text = d->text;
    for (int i = 0; i < 5; i++) x.counts[i] = (len_t)d->counts[i];
    const uint64_t N = d->text_length + 1;
    x.fwd.bv = BitvecIntl4{N, d->bv_fwd, d->cnt_fwd};
    x.fwd.dollarPos = d->dollar_pos_fwd;
    x.rev.bv = BitvecIntl4{N, d->bv_rev, d->cnt_rev};
    x.rev.dollarPos = d->dollar_pos_rev;
    x.bwt.words = d->bwt_words;
    x.bwt.tSize = d->text_length;
    x.saMark = Bitvec9{d->text_length, d->sa_bv, d->sa_bv_counts};
    x.saSamples = d->sa_samples;
    x.sparseness = d->sa_sparseness;
    x.switchPoint = d->switch_point;
    x.wordSize = d->kmer_size;
    x.seqStarts.assign(d->seq_starts, d->seq_starts + d->n_seqs);
    populateTable(x);
    return h;
}
void orc_index_destroy(void* h) { delete (OrcIndex*)h; }

const void* orc_index_kmer_table(void* h, uint64_t* nEntries) {
    Index& x = ((OrcIndex*)h)->idx;
    *nEntries = x.kmerTable.size();
    return x.kmerTable.data(); // 4 x u32 per entry: sa.b, sa.e, rev.b, rev.e
}

// ---- builders (format restatements) ---------------------------------------
void orc_build_bitvec_intl(const uint8_t* codes, uint64_t n, uint64_t* bv, uint64_t* counts,
                           uint64_t* dollarPos) {
    buildBitvecIntl4(codes, n, bv, counts, *dollarPos);
}
void orc_build_bitvec9_counts(const uint64_t* bv, uint64_t nWords, uint64_t* counts) {
    buildBitvec9Counts(bv, nWords, counts);
}
void orc_encode_bwt(const uint8_t* codes, uint64_t n, uint64_t* out) {
    EncodedBWT::encode(codes, n, out);
}
uint64_t orc_bwt_at(const uint64_t* words, uint64_t i) {
    EncodedBWT e;
    e.words = words;
    return e.at(i);
}

// ---- primitive hooks --------------------------------------------------------
// a1: rank
void orc_rank_batch(void* h, int rev, const uint32_t* c, const uint64_t* p, uint64_t n,
                    uint64_t* out) {
    Index& x = ((OrcIndex*)h)->idx;
    const BWTRepr& r = rev ? x.rev : x.fwd;
    for (uint64_t i = 0; i < n; i++) out[i] = r.bv.rank(c[i], p[i]);
}
// a2: occ / cumOcc
void orc_occ_batch(void* h, int rev, const uint32_t* c, const uint64_t* p, uint64_t n,
                   uint64_t* occ, uint64_t* cum) {
    Index& x = ((OrcIndex*)h)->idx;
    const BWTRepr& r = rev ? x.rev : x.fwd;
    for (uint64_t i = 0; i < n; i++) {
        occ[i] = r.occ((int)c[i], p[i]);
        cum[i] = r.cumOcc((int)c[i], p[i]);
    }
}
// a3/a4: all four children of n parents.  mode: 0 forward, 1 backward, 2 uni-backward
// in: n x {sa.b, sa.e, rev.b, rev.e}; out: n x 4 x {sa.b, sa.e, rev.b, rev.e}; ok: n x 4
void orc_extend_batch(void* h, int mode, const uint32_t* in, uint64_t n, uint32_t* out,
                      uint8_t* ok) {
    Index& x = ((OrcIndex*)h)->idx;
    for (uint64_t i = 0; i < n; i++) {
        RangePair p(Range(in[4 * i], in[4 * i + 1]), Range(in[4 * i + 2], in[4 * i + 3]));
        for (len_t c = 1; c < 5; c++) {
            RangePair ch;
            bool o = mode == 0 ? x.extendForward(c, p, ch)
                               : (mode == 1 ? x.extendBackward(c, p, ch) : x.extendBackwardUni(c, p, ch));
            uint32_t* q = out + (i * 4 + (c - 1)) * 4;
            q[0] = ch.sa.b;
            q[1] = ch.sa.e;
            q[2] = ch.rev.b;
            q[3] = ch.rev.e;
            ok[i * 4 + (c - 1)] = o;
        }
    }
}
// a10: locate
void orc_locate_batch(void* h, const uint32_t* rows, uint64_t n, uint32_t* out, uint64_t* lfSteps) {
    Index& x = ((OrcIndex*)h)->idx;
    Counters cnt;
    for (uint64_t i = 0; i < n; i++) out[i] = x.findSA(rows[i], cnt);
    if (lfSteps) *lfSteps = cnt.c[LF_STEPS];
}

// ---- strategy ---------------------------------------------------------------
void* orc_strategy_create(int metric, int partition, uint32_t kmerCutOff) {
    Strategy* s = new Strategy();
    s->metric = (DistanceMetric)metric;
    s->partition = (PartitionStrategy)partition;
    s->useKmerCutOff = kmerCutOff;
    return s;
}
void orc_strategy_destroy(void* s) { delete (Strategy*)s; }
// add one scheme for distance k: nSearches x nParts arrays (row-major)
int orc_strategy_add_scheme(void* sp, uint32_t k, uint32_t nSearches, uint32_t nParts,
                            const uint32_t* pi, const uint32_t* L, const uint32_t* U) {
    Strategy* s = (Strategy*)sp;
    try {
        std::vector<Search> searches;
        for (uint32_t i = 0; i < nSearches; i++) {
            std::vector<len_t> o(pi + i * nParts, pi + (i + 1) * nParts);
            std::vector<len_t> l(L + i * nParts, L + (i + 1) * nParts);
            std::vector<len_t> u(U + i * nParts, U + (i + 1) * nParts);
            searches.push_back(Search::makeSearch(o, l, u, i));
        }
        if (s->schemesPerK.size() <= k) s->schemesPerK.resize(k + 1);
        s->schemesPerK[k].emplace_back(searches, k);
    } catch (const std::exception& e) {
        return -1;
    }
    return 0;
}
void orc_strategy_set_partition_params(void* sp, uint32_t k, const double* seeding, uint32_t nSeed,
                                       const uint64_t* weights, uint32_t nW, const double* begins,
                                       uint32_t nB) {
    Strategy* s = (Strategy*)sp;
    if (s->seedingPositions.size() <= k) s->seedingPositions.resize(k + 1);
    if (s->weights.size() <= k) s->weights.resize(k + 1);
    if (s->begins.size() <= k) s->begins.resize(k + 1);
    s->seedingPositions[k].assign(seeding, seeding + nSeed);
    s->weights[k].assign(weights, weights + nW);
    s->begins[k].assign(begins, begins + nB);
}
// search metadata for pinning against search.h (a14)
// out: directions[n], dswitch[n], low[n], high[n], uniBackwards(idx)[n]
void orc_search_info(const uint32_t* pi, const uint32_t* L, const uint32_t* U, uint32_t n,
                     uint32_t* out) {
    std::vector<len_t> o(pi, pi + n), l(L, L + n), u(U, U + n);
    Search s = Search::makeSearch(o, l, u, 0);
    for (uint32_t i = 0; i < n; i++) {
        out[i] = s.getDirection(i);
        out[n + i] = s.getDirectionSwitch(i);
        out[2 * n + i] = s.lowHigh[i].first;
        out[3 * n + i] = s.lowHigh[i].second;
        out[4 * n + i] = s.isUnidirectionalBackwards(i);
    }
}
uint32_t orc_scheme_critical_part(void* sp, uint32_t k, uint32_t schemeIdx) {
    return ((Strategy*)sp)->schemesPerK[k][schemeIdx].criticalPartIndex;
}

// ---- matrix dump for pinning against bitparallelmatrix.{h,cpp} (a5) ---------
// X: horizontal sequence (already in the order it is accessed), Y: vertical.
// rows_out: per computed row i (1-based, index i-1): {valid, HP, HN, D0, RAC, score,
// at(i,lastCol) if inFinalColumn else ~0, onlyVerticalGapsLeft, firstCol, inFinalColumn}
// returns number of rows computed (stops after the first invalid row)
uint32_t orc_matrix_dump(const char* X, uint32_t xlen, const char* Y, uint32_t ylen,
                         uint32_t maxED, const uint32_t* initED, uint32_t nInit,
                         uint64_t* rows_out, uint32_t* geom /*m,n,Wv,Wh,sizeFinalCol*/) {
    BitParallelED64 M;
    Substring sx(X, xlen, 0, xlen, FORWARD);
    M.setSequence(sx);
    M.initializeMatrix(maxED, std::vector<uint32_t>(initED, initED + nInit));
    geom[0] = M.getNumberOfRows();
    geom[1] = M.getNumberOfCols();
    geom[2] = M.getWv();
    geom[3] = M.getWh();
    geom[4] = M.getSizeOfFinalColumn();
    uint32_t i = 0;
    for (; i < ylen && i + 1 < M.getNumberOfRows(); i++) {
        bool v = M.computeRow(i + 1, Y[i]);
        uint64_t* o = rows_out + (uint64_t)i * 10;
        const BitVectors& r = M.row(i + 1);
        o[0] = v;
        o[1] = r.HP;
        o[2] = r.HN;
        o[3] = r.D0;
        o[4] = r.RAC;
        o[5] = r.score;
        o[6] = M.inFinalColumn(i + 1) ? M.at(i + 1, M.getNumberOfCols() - 1) : ~0ull;
        o[7] = M.onlyVerticalGapsLeft(i + 1);
        o[8] = M.getFirstColumn(i + 1);
        o[9] = M.inFinalColumn(i + 1);
        if (!v) {
            i++;
            break;
        }
    }
    return i;
}

// in-text verification of one pattern against many start positions (a11)
// returns number of occurrences written (<= cap)
uint64_t orc_verify_batch(void* h, const char* pattern, uint32_t plen, const uint32_t* starts,
                          uint64_t n, uint32_t maxED, uint32_t minED, int fixedStart,
                          orc_occ* out, uint64_t cap, uint64_t* counters_out) {
    Index& x = ((OrcIndex*)h)->idx;
    Strategy st;
    Matcher m(x, st);
    std::string pat(pattern, plen);
    Substring p(pat.data(), plen, 0, plen, FORWARD);
    Occurrences occ;
    std::vector<len_t> sp(starts, starts + n);
    m.inTextVerification(sp, maxED, minED, occ, p, fixedStart != 0);
    uint64_t k = 0;
    for (auto& t : occ.inTextOcc) {
        if (k < cap) out[k] = {t.range.b, t.range.e, t.distance, (uint32_t)t.strand};
        k++;
    }
    if (counters_out) memcpy(counters_out, m.counters.c, sizeof(m.counters.c));
    return k;
}

// ---- batch matching (the §8b boundary) -------------------------------------
struct OrcResult {
    std::vector<orc_occ> occs;
    std::vector<uint64_t> offs; // nReads + 1
    Counters counters;
    std::string error;
};

void* orc_match_batch(void* h, void* sp, uint32_t k, const char* seqs, const uint64_t* offs,
                      uint32_t nReads, uint32_t nThreads) {
    Index& x = ((OrcIndex*)h)->idx;
    Strategy& st = *(Strategy*)sp;
    OrcResult* res = new OrcResult();
    std::vector<std::vector<orc_occ>> per(nReads);
    if (nThreads == 0) nThreads = 1;
    std::vector<Counters> cnts(nThreads);
    std::vector<std::string> errs(nThreads);
    std::atomic<uint32_t> next(0);
    auto work = [&](uint32_t tid) {
        Matcher m(x, st);
        try {
            for (;;) {
                uint32_t base = next.fetch_add(64); // processChunk granularity parallel.cpp:67
                if (base >= nReads) break;
                uint32_t end = std::min(nReads, base + 64);
                for (uint32_t r = base; r < end; r++) {
                    std::string read = Matcher::cleanRead(std::string(seqs + offs[r], offs[r + 1] - offs[r]));
                    auto v = m.matchApproxAll(read, k);
                    per[r].reserve(v.size());
                    for (auto& t : v)
                        per[r].push_back({t.range.b, t.range.e, t.distance, (uint32_t)t.strand});
                }
            }
        } catch (const std::exception& e) {
            errs[tid] = e.what();
        }
        cnts[tid] = m.counters;
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < nThreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    for (auto& e : errs)
        if (!e.empty()) res->error = e;
    res->offs.resize(nReads + 1, 0);
    for (uint32_t r = 0; r < nReads; r++) res->offs[r + 1] = res->offs[r] + per[r].size();
    res->occs.reserve(res->offs[nReads]);
    for (uint32_t r = 0; r < nReads; r++)
        res->occs.insert(res->occs.end(), per[r].begin(), per[r].end());
    for (auto& c : cnts) res->counters.add(c);
    return res;
}
// the search view of a b-move index with its k-mer table (populateTable: index loading in the reference, not matching),
// built once per (index, word size)
static std::mutex g_adapterMu;
static std::map<std::pair<const void*, uint32_t>, std::unique_ptr<orc::MoveIndexAdapter>> g_adapters;
static const orc::MoveIndexAdapter& moveAdapter(const orc::BMoveIndex64& bm, uint32_t wordSize) {
    std::lock_guard<std::mutex> lock(g_adapterMu);
    auto& slot = g_adapters[{&bm, wordSize}];
    if (!slot) slot.reset(new orc::MoveIndexAdapter(bm, wordSize));
    return *slot;
}
void orc_move_prepare(void* h, uint32_t wordSize) { (void)moveAdapter(*(orc::BMoveIndex64*)h, wordSize); }
// the same call on the run-length compressed flavour (h: orc_move_create); wordSize: k-mer table of the index
void* orc_move_match_batch(void* h, void* sp, uint32_t k, const char* seqs, const uint64_t* offs, uint32_t nReads,
                           uint32_t nThreads, uint32_t wordSize) {
    const orc::BMoveIndex64& bm = *(orc::BMoveIndex64*)h;
    Strategy& st = *(Strategy*)sp;
    OrcResult* res = new OrcResult();
    std::vector<std::vector<orc_occ>> per(nReads);
    if (nThreads == 0) nThreads = 1;
    std::vector<Counters> cnts(nThreads);
    std::vector<std::string> errs(nThreads);
    std::atomic<uint32_t> next(0);
    try {
        const orc::MoveIndexAdapter& x = moveAdapter(bm, wordSize);
        auto work = [&](uint32_t tid) {
            orc::MoveMatcher m(x, st);
            const uint64_t rows0 = *bm.rowStepsPtr();
            try {
                for (;;) {
                    uint32_t base = next.fetch_add(64);
                    if (base >= nReads) break;
                    uint32_t end = std::min(nReads, base + 64);
                    for (uint32_t r = base; r < end; r++) {
                        std::string read = orc::MoveMatcher::cleanRead(std::string(seqs + offs[r], offs[r + 1] - offs[r]));
                        auto v = m.matchApproxAll(read, k);
                        per[r].reserve(v.size());
                        for (auto& t : v) per[r].push_back({t.range.b, t.range.e, t.distance, (uint32_t)t.strand});
                    }
                }
            } catch (const std::exception& e) {
                errs[tid] = e.what();
            }
            m.counters.inc(ROW_STEPS, *bm.rowStepsPtr() - rows0);
            cnts[tid] = m.counters;
        };
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < nThreads; t++) th.emplace_back(work, t);
        work(0);
        for (auto& t : th) t.join();
    } catch (const std::exception& e) {
        res->error = e.what();
    }
    for (auto& e : errs)
        if (!e.empty()) res->error = e;
    res->offs.resize(nReads + 1, 0);
    for (uint32_t r = 0; r < nReads; r++) res->offs[r + 1] = res->offs[r] + per[r].size();
    res->occs.reserve(res->offs[nReads]);
    for (uint32_t r = 0; r < nReads; r++) res->occs.insert(res->occs.end(), per[r].begin(), per[r].end());
    for (auto& c : cnts) res->counters.add(c);
    return res;
}
// ---- BEST (+x strata) mode for a chunk of reads: occurrences (concatenated-text coordinates) + assignment + CIGAR
struct OrcBest {
    std::vector<orc_occ> occs;
    std::vector<uint32_t> seqId, seqBegin, trimmed;
    std::vector<std::string> cigars;
    std::vector<uint64_t> offs;
    std::vector<uint32_t> best, nHits;
    Counters counters;
    std::string error;
};
extern "C++" {
template <class M, class IX>
static void* matchBestOn(const IX& ix, Strategy& st, uint32_t x, uint32_t minIdentity, uint32_t maxSupported, const char* seqs,
                         const uint64_t* offs, uint32_t nReads, uint32_t nThreads);
}
void* orc_match_best(void* h, void* sp, uint32_t x, uint32_t minIdentity, uint32_t maxSupported, const char* seqs,
                     const uint64_t* offs, uint32_t nReads, uint32_t nThreads) {
    return matchBestOn<Matcher>(((OrcIndex*)h)->idx, *(Strategy*)sp, x, minIdentity, maxSupported, seqs, offs, nReads, nThreads);
}
// the same on the run-length compressed flavour; the text beside the index (CIGARs, trimming: the occurrence's matched string)
void orc_move_attach_text(void* h, uint32_t wordSize, const uint8_t* text, uint64_t n, const uint32_t* seqStarts, uint32_t nStarts) {
    const_cast<orc::MoveIndexAdapter&>(moveAdapter(*(orc::BMoveIndex64*)h, wordSize)).attachText(text, n, seqStarts, nStarts);
}
void* orc_move_match_best(void* h, void* sp, uint32_t x, uint32_t minIdentity, uint32_t maxSupported, const char* seqs,
                          const uint64_t* offs, uint32_t nReads, uint32_t nThreads, uint32_t wordSize) {
    return matchBestOn<orc::MoveMatcher>(moveAdapter(*(orc::BMoveIndex64*)h, wordSize), *(Strategy*)sp, x, minIdentity, maxSupported, seqs, offs,
                                         nReads, nThreads);
}
extern "C++" {
template <class M, class IX>
static void* matchBestOn(const IX& ix, Strategy& st, uint32_t x, uint32_t minIdentity, uint32_t maxSupported, const char* seqs,
                         const uint64_t* offs, uint32_t nReads, uint32_t nThreads) {
    OrcBest* res = new OrcBest();
    std::vector<std::vector<typename M::BestOcc>> per(nReads);
    res->best.assign(nReads, 0xFFFFFFFFu);
    res->nHits.assign(nReads, 0);
    if (nThreads == 0) nThreads = 1;
    std::vector<Counters> cnts(nThreads);
    std::vector<std::string> errs(nThreads);
    std::atomic<uint32_t> next(0);
    auto work = [&](uint32_t tid) {
        M m(ix, st);
        try {
            for (;;) {
                uint32_t base = next.fetch_add(64);
                if (base >= nReads) break;
                uint32_t end = std::min(nReads, base + 64);
                for (uint32_t r = base; r < end; r++) {
                    std::string read = M::cleanRead(std::string(seqs + offs[r], offs[r + 1] - offs[r]));
                    uint32_t best = 0, nHits = 0;
                    bool found = false;
                    per[r] = m.matchApproxBestPlusX(read, x, minIdentity, maxSupported, best, nHits, found);
                    if (found) {
                        res->best[r] = best;
                        res->nHits[r] = nHits;
                    }
                }
            }
        } catch (const std::exception& e) {
            errs[tid] = e.what();
        }
        cnts[tid] = m.counters;
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < nThreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    for (auto& e : errs)
        if (!e.empty()) res->error = e;
    res->offs.resize(nReads + 1, 0);
    for (uint32_t r = 0; r < nReads; r++) {
        res->offs[r + 1] = res->offs[r] + per[r].size();
        for (auto& o : per[r]) {
            res->occs.push_back({o.t.range.b, o.t.range.e, o.t.distance, (uint32_t)o.t.strand});
            res->seqId.push_back(o.seqID);
            res->seqBegin.push_back(o.seqBegin);
            std::string c;
            for (auto& p : o.t.cigar) c += std::to_string(p.second) + p.first;
            res->cigars.push_back(c);
        }
    }
    for (auto& c : cnts) res->counters.add(c);
    return res;
}
} // extern "C++"
const char* orc_best_error(void* r) { return ((OrcBest*)r)->error.c_str(); }
uint64_t orc_best_size(void* r) { return ((OrcBest*)r)->occs.size(); }
void orc_best_copy(void* r, orc_occ* occs, uint32_t* seqId, uint32_t* seqBegin, uint64_t* offs, uint32_t* best,
                   uint32_t* nHits, uint64_t* counters) {
    OrcBest* res = (OrcBest*)r;
    if (!res->occs.empty()) {
        memcpy(occs, res->occs.data(), res->occs.size() * sizeof(orc_occ));
        memcpy(seqId, res->seqId.data(), res->seqId.size() * 4);
        memcpy(seqBegin, res->seqBegin.data(), res->seqBegin.size() * 4);
    }
    memcpy(offs, res->offs.data(), res->offs.size() * 8);
    if (!res->best.empty()) {
        memcpy(best, res->best.data(), res->best.size() * 4);
        memcpy(nHits, res->nHits.data(), res->nHits.size() * 4);
    }
    if (counters) memcpy(counters, res->counters.c, sizeof(res->counters.c));
}
const char* orc_best_cigar(void* r, uint64_t i) { return ((OrcBest*)r)->cigars[i].c_str(); }
void orc_best_free(void* r) { delete (OrcBest*)r; }

// ---- SAM text of a chunk of reads in ALL mode (ids: "\n"-joined read identifiers as they stand in the FASTQ, quals likewise)
struct OrcText {
    std::string text, error;
};
void* orc_match_batch_sam(void* h, void* sp, uint32_t k, const char* seqs, const uint64_t* offs, uint32_t nReads,
                          const char* ids, const char* quals, const char* seqNames, int unmapped, int xa) {
    Index& ix = ((OrcIndex*)h)->idx;
    Strategy& st = *(Strategy*)sp;
    OrcText* res = new OrcText();
    auto split = [](const char* s2) {
        std::vector<std::string> v;
        std::string cur;
        for (const char* p = s2; *p; p++)
            if (*p == '\n') {
                v.push_back(cur);
                cur.clear();
            } else cur += *p;
        v.push_back(cur);
        return v;
    };
    const std::vector<std::string> vid = split(ids), vq = split(quals), names = split(seqNames);
    try {
        Matcher m(ix, st);
        for (uint32_t r = 0; r < nReads; r++) {
            std::string read = Matcher::cleanRead(std::string(seqs + offs[r], offs[r + 1] - offs[r]));
            res->text += m.samRecordsAll(read, k, cleanSeqID(vid[r]), vq[r], names, unmapped != 0, xa != 0);
        }
    } catch (const std::exception& e) {
        res->error = e.what();
    }
    return res;
}
const char* orc_text_get(void* r) { return ((OrcText*)r)->text.c_str(); }
const char* orc_text_error(void* r) { return ((OrcText*)r)->error.c_str(); }
void orc_text_free(void* r) { delete (OrcText*)r; }

const char* orc_result_error(void* r) { return ((OrcResult*)r)->error.c_str(); }
uint64_t orc_result_size(void* r) { return ((OrcResult*)r)->occs.size(); }
void orc_result_copy(void* r, orc_occ* occs, uint64_t* offs, uint64_t* counters) {
    OrcResult* res = (OrcResult*)r;
    if (occs && !res->occs.empty()) memcpy(occs, res->occs.data(), res->occs.size() * sizeof(orc_occ));
    if (offs) memcpy(offs, res->offs.data(), res->offs.size() * 8);
    if (counters) memcpy(counters, res->counters.c, sizeof(res->counters.c));
}
void orc_result_free(void* r) { delete (OrcResult*)r; }
uint32_t orc_num_counters() { return COUNTER_TYPE_MAX; }

// ---- run-length compressed backend (oracle_move.hpp) -----------------------------------------------------------------
// same record as include/columba_amd.h: cmb_move_range
struct orc_move_range {
    uint64_t begin, end, begin_run, end_run;
    uint64_t rev_begin, rev_end, rev_begin_run, rev_end_run;
    uint64_t toehold;
    uint32_t original_depth;
    uint8_t runs_valid, rev_runs_valid, toehold_represents_end, reserved;
};
static orc::MovePair64 toPair(const orc_move_range& r) {
    return orc::MovePair64(orc::MoveRange64(r.begin, r.end, r.begin_run, r.end_run, r.runs_valid != 0),
                           orc::MoveRange64(r.rev_begin, r.rev_end, r.rev_begin_run, r.rev_end_run, r.rev_runs_valid != 0), r.toehold,
                           r.toehold_represents_end != 0, r.original_depth);
}
static orc_move_range fromPair(const orc::MovePair64& p) {
    orc_move_range r;
    r.begin = p.sa.begin, r.end = p.sa.end, r.begin_run = p.sa.beginRun, r.end_run = p.sa.endRun;
    r.rev_begin = p.rev.begin, r.rev_end = p.rev.end, r.rev_begin_run = p.rev.beginRun, r.rev_end_run = p.rev.endRun;
    r.toehold = p.toehold, r.original_depth = (uint32_t)p.originalDepth;
    r.runs_valid = p.sa.runIndicesValid, r.rev_runs_valid = p.rev.runIndicesValid, r.toehold_represents_end = p.toeholdRepresentsEnd;
    r.reserved = 0;
    return r;
}

void* orc_move_create(const uint8_t* lf, uint64_t lfLen, const uint8_t* lr, uint64_t lrLen, const uint64_t* smpf, const uint64_t* smpl,
                      const uint64_t* rsmpf, const uint64_t* rsmpl, const uint64_t* predFirst, const uint64_t* firstToRun,
                      const uint64_t* predLast, const uint64_t* lastToRun, const uint32_t* plcp) {
    auto* ix = new orc::BMoveIndex64();
    if (!ix->move.loadBytes(lf, lfLen) || !ix->moveR.loadBytes(lr, lrLen)) {
        delete ix;
        return nullptr;
    }
    ix->textLength = ix->move.getTextSize();
    const uint64_t r = ix->move.size(), rr = ix->moveR.size();
    ix->samplesFirst.assign(smpf, smpf + r);
    ix->samplesLast.assign(smpl, smpl + r);
    ix->revSamplesFirst.assign(rsmpf, rsmpf + rr);
    ix->revSamplesLast.assign(rsmpl, rsmpl + rr);
    if (predFirst) {
        ix->predFirst.assign(predFirst, predFirst + r);
        ix->firstToRun.assign(firstToRun, firstToRun + r);
        ix->predLast.assign(predLast, predLast + r);
        ix->lastToRun.assign(lastToRun, lastToRun + r);
        ix->plcp.assign(plcp, plcp + ix->textLength);
    }
    return ix;
}
void orc_move_destroy(void* h) {
    {
        std::lock_guard<std::mutex> lock(g_adapterMu);
        for (auto it = g_adapters.begin(); it != g_adapters.end();) it = it->first.first == h ? g_adapters.erase(it) : std::next(it);
    }
    delete (orc::BMoveIndex64*)h;
}
void orc_move_complete_range(void* h, orc_move_range* out) { *out = fromPair(((orc::BMoveIndex64*)h)->getCompleteRange()); }
// rows of a table as (head, inputStart, outputStart, outputRun), nrOfRuns + 1 of them; returns nrOfRuns
uint64_t orc_move_rows(void* h, int rev, uint64_t* out) {
    auto* ix = (orc::BMoveIndex64*)h;
    const auto& m = rev ? ix->moveR : ix->move;
    if (out)
        for (uint64_t i = 0; i <= m.size(); i++) {
            out[4 * i] = m.getRunHead(i), out[4 * i + 1] = m.getInputStartPos(i);
            out[4 * i + 2] = m.getOutputStartPos(i), out[4 * i + 3] = m.getOutputStartRun(i);
        }
    return m.size();
}
// mode 0 forward, 1 backward, 2 unidirectional backward (as cmb_extend_batch); c = 1..4; returns the rows stepped over
uint64_t orc_move_extend(void* h, int mode, uint64_t n, const orc_move_range* parents, const uint8_t* c, orc_move_range* children,
                         uint8_t* ok) {
    auto* ix = (orc::BMoveIndex64*)h;
    const uint64_t before = *ix->rowStepsPtr();
    for (uint64_t i = 0; i < n; i++) {
        orc::MovePair64 child;
        ok[i] = ix->extend(mode, c[i], toPair(parents[i]), child);
        children[i] = fromPair(child);
    }
    return *ix->rowStepsPtr() - before;
}
// text positions of a range (bmove.cpp:543-560); returns the count (positions beyond cap are not stored)
uint64_t orc_move_locate(void* h, const orc_move_range* r, uint64_t* out, uint64_t cap) {
    auto* ix = (orc::BMoveIndex64*)h;
    std::vector<uint64_t> pos;
    ix->locate(toPair(*r), pos);
    for (uint64_t i = 0; i < pos.size() && i < cap; i++) out[i] = pos[i];
    return pos.size();
}

// k = 0 on the b-move index: matchApproxAllMap with maxED = 0 (searchstrategy.cpp:499-510): exactMatchesOutput of the read
// (forward strand) and of its reverse complement.  occ: {begin, end, distance = 0, strand} as 4 x uint64 per occurrence, in
// the reference's order; offsets: n + 1.  counters[0] NODE_COUNTER, counters[1] TOTAL_REPORTED_POSITIONS.  Returns the
// number of occurrences (those beyond cap are not stored).
uint64_t orc_move_match_exact(void* h, const char* reads, const uint64_t* readOffsets, uint64_t n, uint64_t* occ, uint64_t cap,
                              uint64_t* offsets, uint64_t* counters) {
    auto* ix = (orc::BMoveIndex64*)h;
    uint64_t total = 0, nodes = 0;
    auto code = [](char c) { // reads.h:43-58 (upper case; everything outside ACGT becomes N), alphabet.h c2i
        switch (c) {
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 3;
        case 'T': case 't': return 4;
        default: return -1;
        }
    };
    for (uint64_t i = 0; i < n; i++) {
        offsets[i] = total;
        const uint64_t len = readOffsets[i + 1] - readOffsets[i];
        std::vector<int> fw(len), rc(len);
        for (uint64_t j = 0; j < len; j++) {
            fw[j] = code(reads[readOffsets[i] + j]);
            rc[len - 1 - j] = fw[j] < 0 ? -1 : 5 - fw[j]; // nucleotide.h:250 (reverse complement; N stays N)
        }
        for (int strand = 0; strand < 2; strand++) {
            std::vector<uint64_t> pos;
            ix->exactMatches(strand ? rc : fw, pos, nodes);
            for (uint64_t p : pos) {
                if (total < cap) occ[4 * total] = p, occ[4 * total + 1] = p + len, occ[4 * total + 2] = 0, occ[4 * total + 3] = strand;
                total++;
            }
        }
    }
    offsets[n] = total;
    counters[0] = nodes;
    counters[1] = total;
    return total;
}

// populateTable of the RLC flavour: 4^wordSize records
void orc_move_kmer_table(void* h, uint32_t wordSize, orc_move_range* out) {
    const auto t = ((orc::BMoveIndex64*)h)->kmerTable(wordSize);
    for (size_t i = 0; i < t.size(); i++) out[i] = fromPair(t[i]);
}

// the 32-bit narrow-block experiment: {phases on the 32-bit matrix, phases that fell back}; reset on request
void orc_narrow32_stats(uint64_t* out, int reset) {
    for (int i = 0; i < 2; i++) {
        out[i] = orc::g_narrow32Stats[i].load();
        if (reset) orc::g_narrow32Stats[i].store(0);
    }
}

} // extern "C"

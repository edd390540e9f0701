// columba_amd.hpp — C++ host adapter over the C-ABI (include/columba_amd.h).
//
// Keeps the names, argument meaning and error behaviour of the reference's host API for the hot
// path, so that Columba-style driver code (processChunk, src/parallel.cpp:67-78) ports by changing
// the namespace:
//
//   reference (src/)                                         here (namespace columba_amd)
//   -------------------------------------------------------  -------------------------------------
//   FMIndex(base, inTextSwitch, noCIGAR, sa_sparse,           FMIndex(base, inTextSwitch, noCIGAR,
//           verbose, wordSize)        fmindex/fmindex.h:403           sa_sparse, verbose, wordSize)
//   KucherovKPlus1 / PigeonHoleSearchStrategy /               same class names (thin subclasses of
//   MultipleSchemesStrategy / CustomSearchStrategy            SearchStrategy)
//                         searchstrategy.h:2829,3221,2584,2130
//   SearchStrategy::matchApprox(ReadBundle&, maxED,           same signature, plus matchApproxBatch
//           Counters&, std::vector<TextOcc>&)   :2021-2024    for a whole chunk (one GPU batch)
//   Counters / TextOcc / Range / ReadBundle                   value types with the same accessors
//                         indexhelpers.h:1846,289,63; reads.h:128
//
// Errors: every failure of the C-ABI is re-thrown as std::runtime_error, which is what the
// reference throws from its loaders and parsers (fmindex.cpp:84, indexinterface.cpp:100,
// search.h:559-586) and what its main() catches (parallel.cpp:1041-1044).
#pragma once
#include "columba_amd.h"
#include "columba_amd_best.hpp"

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <map>
#include <string>
#include <vector>

namespace columba_amd {

typedef uint32_t length_t;
enum Strand { FORWARD_STRAND = 0, REVERSE_C_STRAND = 1 };
enum PartitionStrategy { UNIFORM = CMB_PARTITION_UNIFORM, STATIC = CMB_PARTITION_STATIC, DYNAMIC = CMB_PARTITION_DYNAMIC };
enum DistanceMetric { HAMMING = CMB_METRIC_HAMMING, EDIT = CMB_METRIC_EDIT };

inline void check(int rc) {
    if (rc != CMB_OK) throw std::runtime_error(cmb_last_error());
}

class Range {
    length_t b, e;

  public:
    Range(length_t begin = 0, length_t end = 0) : b(begin), e(end) {}
    length_t getBegin() const { return b; }
    length_t getEnd() const { return e; }
    length_t width() const { return e > b ? e - b : 0; }
};

class TextOcc {
    Range range;
    length_t distance;
    Strand strand;

  public:
    TextOcc(Range r, length_t d, Strand s) : range(r), distance(d), strand(s) {}
    const Range& getRange() const { return range; }
    length_t getBegin() const { return range.getBegin(); }
    length_t getEnd() const { return range.getEnd(); }
    length_t getDistance() const { return distance; }
    Strand getStrand() const { return strand; }
    bool isRevCompl() const { return strand == REVERSE_C_STRAND; }
};

class Counters {
  public:
    enum CounterType {
        NODE_COUNTER = CMB_CNT_NODE,
        TOTAL_REPORTED_POSITIONS = CMB_CNT_TOTAL_REPORTED,
        IN_TEXT_STARTED = CMB_CNT_IN_TEXT_STARTED,
        ABORTED_IN_TEXT_VERIF = CMB_CNT_ABORTED_IN_TEXT,
        CIGARS_IN_TEXT_VERIFICATION = CMB_CNT_CIGARS_IN_TEXT,
        IMMEDIATE_SWITCH = CMB_CNT_IMMEDIATE_SWITCH,
        SEARCH_STARTED = CMB_CNT_SEARCH_STARTED,
        COUNTER_TYPE_MAX = CMB_CNT_MAX
    };
    Counters() { counters.fill(0); }
    void resetCounters() { counters.fill(0); }
    void inc(CounterType t, uint64_t amount = 1) { counters[t] += amount; }
    uint64_t get(CounterType t) const { return counters[t]; }
    void addRaw(const uint64_t* c) {
        for (int i = 0; i < CMB_CNT_MAX; i++) counters[i] += c[i];
    }

  private:
    std::array<uint64_t, CMB_CNT_MAX> counters;
};

class ReadBundle { // reads.h:128 (clean-up and reverse complement happen on the device)
    std::string seqID, read;

  public:
    ReadBundle(const std::string& id, const std::string& sequence) : seqID(id), read(sequence) {}
    const std::string& getSeqID() const { return seqID; }
    const std::string& getRead() const { return read; }
    size_t size() const { return read.size(); }
};

class FMIndex {
    cmb_index* h = nullptr;
    length_t textLength = 0, switchPoint = 0;
    std::vector<length_t> startPos;
    std::vector<std::string> seqNames;

    template <typename T> static std::vector<T> slurp(std::ifstream& f, size_t n) {
        std::vector<T> v(n);
        f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(T)));
        if (!f) throw std::runtime_error("Problem reading index file (truncated)");
        return v;
    }

  public:
    // Loads the Vanilla index files written by columba_build (formats: SURVEY.md §5) and uploads them.
    FMIndex(const std::string& baseFile, length_t inTextSwitch, bool /*noCIGAR*/, int sa_sparse = 1,
            bool /*verbose*/ = true, length_t wordSize = 10, int device = 0)
        : switchPoint(inTextSwitch) {
        { // indexinterface.cpp:77-128
            std::ifstream m(baseFile + ".meta");
            if (m) {
                length_t tag;
                size_t sz;
                std::string flavour;
                m >> tag >> sz >> flavour;
                if (sz != sizeof(length_t))
                    throw std::runtime_error("The index was built with a compiled version that uses " +
                                             std::to_string(sz * 8) + "-bit numbers, while the current programme was "
                                             "compiled using 32-bit numbers. Recompile the programme with the correct "
                                             "THIRTY_TWO flag set or rebuild the index.");
                if (flavour != "VANILLA")
                    throw std::runtime_error("The index was built with a different flavor of Columba.");
            }
        }
        std::ifstream cct(baseFile + ".cct", std::ios::binary);
        if (!cct) throw std::runtime_error("Cannot open file: " + baseFile + ".cct");
        auto charCounts = slurp<length_t>(cct, 256);
        cmb_index_desc d{};
        uint64_t cum = 0;
        int nsym = 0;
        for (size_t i = 0; i < 256; i++) {
            if (!charCounts[i]) continue;
            if (nsym < 5) d.counts[nsym] = cum;
            nsym++;
            cum += charCounts[i];
        }
        if (nsym != 5) throw std::runtime_error("The index alphabet must be $ACGT");
        std::ifstream txt(baseFile + ".txt.bin", std::ios::binary);
        if (!txt) throw std::runtime_error("Error opening file for reading: " + baseFile + ".txt.bin");
        length_t n;
        txt.read(reinterpret_cast<char*>(&n), sizeof(n));
        auto text = slurp<uint8_t>(txt, n);
        textLength = n;
        auto readBrt = [&](const std::string& fn, uint64_t& dollar, std::vector<uint64_t>& bv, std::vector<uint64_t>& cnt) {
            std::ifstream f(fn, std::ios::binary);
            if (!f) throw std::runtime_error("Cannot open file: " + fn);
            uint64_t N;
            f.read(reinterpret_cast<char*>(&dollar), 8);
            f.read(reinterpret_cast<char*>(&N), 8);
            bv = slurp<uint64_t>(f, 4 * ((N + 63) / 64));
            cnt = slurp<uint64_t>(f, 8 * ((N + 511) / 512));
        };
        std::vector<uint64_t> bvF, cF, bvR, cR;
        readBrt(baseFile + ".brt", d.dollar_pos_fwd, bvF, cF);
        readBrt(baseFile + ".rev.brt", d.dollar_pos_rev, bvR, cR);
        const std::string sfx = std::to_string(sa_sparse);
        std::ifstream sab(baseFile + ".sa.bv." + sfx, std::ios::binary);
        if (!sab)
            throw std::runtime_error("Cannot open file: " + baseFile + ".sa.bv." + sfx +
                                     ". Did you set an incorrect suffix array sparseness factor using the -s flag "
                                     "or move your index files?");
        uint64_t N;
        sab.read(reinterpret_cast<char*>(&N), 8);
        const uint64_t nw = (N + 63) / 64;
        auto saBv = slurp<uint64_t>(sab, nw);
        auto saCnt = slurp<uint64_t>(sab, (nw + 7) / 4);
        std::ifstream sas(baseFile + ".sa." + sfx, std::ios::binary | std::ios::ate);
        if (!sas) throw std::runtime_error("Problem reading file: " + baseFile + ".sa." + sfx);
        const size_t nSamples = (size_t)sas.tellg() / sizeof(length_t);
        sas.seekg(0);
        auto samples = slurp<length_t>(sas, nSamples);
        std::ifstream pos(baseFile + ".pos", std::ios::binary | std::ios::ate);
        if (!pos) throw std::runtime_error("Cannot open file: " + baseFile + ".pos\nIs the reference index outdated?");
        const size_t np = (size_t)pos.tellg() / sizeof(length_t);
        pos.seekg(0);
        startPos = slurp<length_t>(pos, np);
        d.text_length = n;
        d.text = text.data();
        d.bv_fwd = bvF.data();
        d.cnt_fwd = cF.data();
        d.bv_rev = bvR.data();
        d.cnt_rev = cR.data();
        d.sa_bv = saBv.data();
        d.sa_bv_counts = saCnt.data();
        d.sa_samples = samples.data();
        d.n_samples = nSamples;
        d.sa_sparseness = (uint32_t)sa_sparse;
        d.seq_starts = startPos.data();
        d.n_seqs = (uint32_t)startPos.size();
        d.kmer_size = wordSize;
        d.in_text_switch = inTextSwitch;
        check(cmb_index_create(&d, device, &h));
    }
    FMIndex(const FMIndex&) = delete;
    FMIndex& operator=(const FMIndex&) = delete;
    ~FMIndex() { cmb_index_destroy(h); }
    cmb_index* handle() const { return h; }
    length_t getSwitchPoint() const { return switchPoint; }
    length_t getTextLength() const { return textLength; }
    const std::vector<length_t>& getStartPositions() const { return startPos; }
};

class SearchStrategy {
  protected:
    FMIndex& index;
    cmb_strategy* h = nullptr;
    SearchStrategy(FMIndex& idx) : index(idx) {}

  public:
    SearchStrategy(const SearchStrategy&) = delete;
    virtual ~SearchStrategy() { cmb_strategy_destroy(h); }
    // reads of the last matchApproxBatch chunk that were matched by naive backtracking instead of a search scheme (not longer than
    // the number of parts: searchstrategy.cpp:148-152) — information only, their occurrences are in the result like all others
    std::vector<size_t> matchedNaively;

    // the body of processChunk's loop for a whole chunk: result[i] = occurrences of reads[i]
    void matchApproxBatch(const std::vector<ReadBundle>& reads, length_t maxED, Counters& counters,
                          std::vector<std::vector<TextOcc>>& result) {
        std::string seqs;
        std::vector<uint64_t> offs(reads.size() + 1, 0);
        for (size_t i = 0; i < reads.size(); i++) {
            seqs += reads[i].getRead();
            offs[i + 1] = seqs.size();
        }
        cmb_batch* b = nullptr;
        check(cmb_batch_create(index.handle(), h, maxED, seqs.data(), offs.data(), (uint32_t)reads.size(), &b));
        struct Guard {
            cmb_batch* b;
            ~Guard() { cmb_batch_destroy(b); }
        } guard{b};
        check(cmb_batch_run(b));
        matchedNaively.clear();
        {
            std::vector<uint8_t> status(reads.size() ? reads.size() : 1);
            uint32_t flagged = 0;
            check(cmb_batch_read_status(b, status.data(), &flagged));
            for (size_t i = 0; flagged && i < reads.size(); i++)
                if (status[i] & CMB_READ_NAIVE_FALLBACK) matchedNaively.push_back(i);
        }
        uint64_t n = 0;
        check(cmb_batch_result_size(b, &n));
        std::vector<cmb_occ> occ(n ? n : 1);
        std::vector<uint64_t> oo(reads.size() + 1), cnt(CMB_CNT_MAX);
        check(cmb_batch_results(b, occ.data(), occ.size(), oo.data(), cnt.data()));
        counters.addRaw(cnt.data());
        result.assign(reads.size(), {});
        for (size_t i = 0; i < reads.size(); i++)
            for (uint64_t j = oo[i]; j < oo[i + 1]; j++)
                result[i].emplace_back(Range(occ[j].begin, occ[j].end), occ[j].distance, (Strand)occ[j].strand);
    }
    // the SAM text of a chunk in ALL mode (matchApproxAllMap + generateOutputSingleEnd, searchstrategy.cpp:495-535, :1824-1902)
    std::string samOfChunkAll(const std::string& seqs, const std::vector<uint64_t>& offs, const std::vector<const char*>& ids,
                              const std::vector<const char*>& quals, const std::vector<const char*>& seqNames, length_t maxED,
                              bool unmappedRecords, bool xaTag) {
        cmb_batch* b = nullptr;
        check(cmb_batch_create(index.handle(), h, maxED, seqs.data(), offs.data(), (uint32_t)(offs.size() - 1), &b));
        struct Guard {
            cmb_batch* b;
            ~Guard() { cmb_batch_destroy(b); }
        } guard{b};
        check(cmb_batch_want_alignments(b, 1));
        check(cmb_batch_run(b));
        const int64_t n = cmb_batch_sam(b, seqs.data(), ids.data(), quals.data(), seqNames.data(), unmappedRecords, xaTag, nullptr, 0);
        if (n < 0) check((int)n);
        std::string text((size_t)n + 1, '\0');
        cmb_batch_sam(b, seqs.data(), ids.data(), quals.data(), seqNames.data(), unmappedRecords, xaTag, &text[0], (uint64_t)n + 1);
        text.resize((size_t)n);
        return text;
    }
    // ---- read pairs in ALL mode
    // the occurrences of one read as cmb_pair_sam takes them: sequence assigned (trimmed where they ran over its end, findSeqName,
    // indexinterface.cpp:833-899; those for which that fails are dropped), with their CIGAR operations
    struct PairOccStore {
        cmb_pair_occ p;
        std::vector<uint16_t> ops;
    };
    typedef std::vector<std::vector<PairOccStore>> MateLists; // [read][occurrence]
    // one batch over recs[ids]: ALL mode at maxED with alignments; perStrand: every strand filtered by itself (mapRead, searchstrategy.h:490-519)
    // instead of the read's two strands together (matchApproxAllMap)
    template <class Record> MateLists listsOfMate(const std::vector<Record>& recs, const std::vector<size_t>& ids, length_t maxED, bool perStrand) {
        std::string seqs;
        std::vector<uint64_t> offs(ids.size() + 1, 0);
        for (size_t j = 0; j < ids.size(); j++) seqs += recs[ids[j]].read, offs[j + 1] = seqs.size();
        cmb_batch* b = nullptr;
        check(cmb_batch_create(index.handle(), h, maxED, seqs.data(), offs.data(), (uint32_t)ids.size(), &b));
        struct Guard {
            cmb_batch* b;
            ~Guard() { cmb_batch_destroy(b); }
        } guard{b};
        check(cmb_batch_want_alignments(b, 1));
        check(cmb_batch_filter_per_strand(b, perStrand ? 1 : 0));
        check(cmb_batch_run(b));
        uint64_t n = 0, nOps = 0;
        check(cmb_batch_result_size(b, &n));
        std::vector<cmb_occ> occ(n ? n : 1);
        std::vector<cmb_aln> aln(n ? n : 1);
        std::vector<uint64_t> oo(ids.size() + 1), cnt(CMB_CNT_MAX);
        check(cmb_batch_results(b, occ.data(), occ.size(), oo.data(), cnt.data()));
        (void)cmb_batch_alignments(b, aln.data(), 0, nullptr, 0, &nOps); // sizes first
        std::vector<uint16_t> ops(nOps ? nOps : 1);
        check(cmb_batch_alignments(b, aln.data(), aln.size(), ops.data(), ops.size(), &nOps));
        MateLists out(ids.size());
        std::vector<char> id, sq, rc, rq;
        std::vector<uint16_t> trimmed(2 * (size_t)maxED + 8);
        for (size_t j = 0; j < ids.size(); j++) {
            const Record& r = recs[ids[j]];
            bool prepared = false;
            for (uint64_t q = oo[j]; q < oo[j + 1]; q++) {
                cmb_occ o = occ[q];
                cmb_aln a = aln[q];
                const uint16_t* src = ops.data() + a.cigar_off;
                uint32_t nOpsQ = a.cigar_len;
                if (a.spans == 1) { // runs past the end of its sequence: assignSequence -> findSeqName trims it
                    if (!prepared) {
                        id.resize(r.seqID.size() + 1), sq.resize(r.read.size() + 1), rc.resize(r.read.size() + 1), rq.resize(r.qual.size() + 1);
                        check(cmb_read_prepare(r.seqID.c_str(), r.read.c_str(), r.qual.c_str(), id.data(), sq.data(), rc.data(), rq.data()));
                        prepared = true;
                    }
                    int found = 0;
                    const char* pat = o.strand ? rc.data() : sq.data();
                    check(cmb_trim_occurrence(index.handle(), pat, (uint32_t)strlen(pat), maxED, CMB_METRIC_EDIT, &o, &a, trimmed.data(), (uint32_t)trimmed.size(),
                                              &nOpsQ, &found));
                    if (!found) continue; // NOT_FOUND: takes no part in the pairing
                    src = trimmed.data();
                }
                PairOccStore st;
                st.ops.assign(src, src + nOpsQ);
                st.p.seq_id = a.seq_id, st.p.begin = a.seq_begin, st.p.end = a.seq_begin + (o.end - o.begin), st.p.index_begin = o.begin;
                st.p.distance = o.distance, st.p.strand = o.strand, st.p.cigar_ops = nullptr, st.p.n_ops = nOpsQ;
                out[j].push_back(std::move(st));
            }
        }
        return out;
    }
    // cmb_pair_sam per pair over the mates' lists (pairSingleEndedMatchesAll + generateSAMPairedEnd)
    template <class Record>
    std::string pairListsAll(const std::vector<Record>& mates1, const std::vector<Record>& mates2, MateLists& L1, MateLists& L2, const std::vector<const char*>& seqNames,
                             uint32_t orientation, uint32_t maxFragSize, uint32_t minFragSize, bool discordantAllowed, bool unmappedRecords, size_t& mappedPairs) {
        const cmb_pair_params prm = {orientation, maxFragSize, minFragSize, discordantAllowed ? 1 : 0, unmappedRecords ? 1 : 0};
        const std::vector<Record>* in[2] = {&mates1, &mates2};
        MateLists* L[2] = {&L1, &L2};
        std::string text;
        std::vector<char> buf;
        for (size_t i = 0; i < mates1.size(); i++) {
            std::vector<cmb_pair_occ> po[2];
            std::vector<char> id[2], sq[2], rc[2], rq[2];
            cmb_pair_read rd[2];
            for (int m = 0; m < 2; m++) {
                const Record& r = (*in[m])[i];
                id[m].resize(r.seqID.size() + 1), sq[m].resize(r.read.size() + 1), rc[m].resize(r.read.size() + 1), rq[m].resize(r.qual.size() + 1);
                check(cmb_read_prepare(r.seqID.c_str(), r.read.c_str(), r.qual.c_str(), id[m].data(), sq[m].data(), rc[m].data(), rq[m].data()));
                for (PairOccStore& st : (*L[m])[i]) {
                    st.p.cigar_ops = st.ops.data();
                    po[m].push_back(st.p);
                }
                rd[m] = cmb_pair_read{id[m].data(), sq[m].data(), rc[m].data(), r.qual.c_str(), rq[m].data(), po[m].data(), (uint32_t)po[m].size()};
            }
            uint32_t nPairs = 0;
            const int64_t n = cmb_pair_sam(&prm, &rd[0], &rd[1], seqNames.data(), nullptr, 0, &nPairs);
            if (n < 0) check((int)n);
            buf.resize((size_t)n + 1);
            cmb_pair_sam(&prm, &rd[0], &rd[1], seqNames.data(), buf.data(), (uint64_t)n + 1, &nPairs);
            text.append(buf.data(), (size_t)n);
            mappedPairs += nPairs > 0;
        }
        return text;
    }
    // The single-end phase that infers the paired-end parameters in ALL mode (parallel.cpp:236-312, :700-727; cmb_pair_infer for :329-466): read 1 of
    // every pair matched as a single read (its two strands filtered together, matchApproxAllMap); read 2 where read 1 has exactly one match in
    // the first reference file; the pairs whose mates both do are the sample.  The lists stay: the chunk is then paired from them
    // (pairSingleEndedMatchesAll, searchstrategy.cpp:1345-1399 — a read 2 that was not matched yet is matched strand by strand there).
    struct PairedEndInferenceAll {
        cmb_pair_inferred inferred{};
        size_t readsGiven = 0, unambiguousPairs = 0;
        MateLists first, second;
        std::vector<uint8_t> read2done;
    };
    template <class Record>
    PairedEndInferenceAll inferPairedEndParametersAll(const std::vector<Record>& mates1, const std::vector<Record>& mates2, length_t maxED,
                                                      uint32_t seqsInFirstFile) {
        if (mates1.size() != mates2.size()) throw std::runtime_error("the two read files do not hold the same number of reads");
        const size_t n = mates1.size();
        PairedEndInferenceAll inf;
        inf.read2done.assign(n, 0);
        std::vector<size_t> all(n), second;
        for (size_t i = 0; i < n; i++) all[i] = i;
        inf.first = listsOfMate(mates1, all, maxED, false);
        auto unambiguous = [&](std::vector<PairOccStore>& v) { // hasUnambiguousMatchInFirstFile: the one match of the first file moves to the front
            size_t count = 0, at = 0;
            for (size_t j = 0; j < v.size(); j++)
                if (v[j].p.seq_id < seqsInFirstFile) {
                    if (++count == 2) return false;
                    at = j;
                }
            if (count == 1) std::swap(v[0], v[at]);
            return count == 1;
        };
        for (size_t i = 0; i < n; i++)
            if (unambiguous(inf.first[i])) second.push_back(i), inf.read2done[i] = 1;
        MateLists done = listsOfMate(mates2, second, maxED, false);
        inf.second.assign(n, {});
        std::vector<cmb_pair_sample> samples;
        for (size_t j = 0; j < second.size(); j++) {
            const size_t i = second[j];
            inf.second[i] = std::move(done[j]);
            if (unambiguous(inf.second[i])) {
                const cmb_pair_occ &a = inf.first[i][0].p, &b = inf.second[i][0].p;
                samples.push_back(cmb_pair_sample{a.begin, a.end, a.strand, b.begin, b.end, b.strand});
            }
        }
        inf.readsGiven = 2 * n, inf.unambiguousPairs = samples.size();
        check(cmb_pair_infer(samples.data(), samples.size(), &inf.inferred));
        return inf;
    }
    // the SAM text of a chunk of read PAIRS in ALL mode: both mates matched single-ended (one batch each, every strand filtered by itself as
    // matchApproxPairedEndAll's mapRead does, searchstrategy.cpp:746-776), then paired as SearchStrategy::pairSingleEndedMatchesAll does
    // (searchstrategy.cpp:1345-1399) with the records of generateSAMPairedEnd.  orientation: CMB_ORIENTATION_*.  startFrom: the lists of the
    // inference phase (read 1, and read 2 where it was matched); the other reads 2 are matched here, strand by strand.
    template <class Record>
    std::string samOfChunkPairedAll(const std::vector<Record>& mates1, const std::vector<Record>& mates2,
                                    const std::vector<const char*>& seqNames, length_t maxED, uint32_t orientation, uint32_t maxFragSize,
                                    uint32_t minFragSize, bool discordantAllowed, bool unmappedRecords, size_t& mappedPairs,
                                    PairedEndInferenceAll* startFrom = nullptr) {
        if (mates1.size() != mates2.size()) throw std::runtime_error("the two read files do not hold the same number of reads");
        const size_t n = mates1.size();
        std::vector<size_t> all(n);
        for (size_t i = 0; i < n; i++) all[i] = i;
        if (!startFrom) {
            MateLists L1 = listsOfMate(mates1, all, maxED, true), L2 = listsOfMate(mates2, all, maxED, true);
            return pairListsAll(mates1, mates2, L1, L2, seqNames, orientation, maxFragSize, minFragSize, discordantAllowed, unmappedRecords, mappedPairs);
        }
        if (startFrom->read2done.size() != n) throw std::runtime_error("the single-end results are those of another chunk");
        std::vector<size_t> rest;
        for (size_t i = 0; i < n; i++)
            if (!startFrom->read2done[i]) rest.push_back(i);
        MateLists late = listsOfMate(mates2, rest, maxED, true);
        for (size_t j = 0; j < rest.size(); j++) startFrom->second[rest[j]] = std::move(late[j]);
        return pairListsAll(mates1, mates2, startFrom->first, startFrom->second, seqNames, orientation, maxFragSize, minFragSize, discordantAllowed,
                            unmappedRecords, mappedPairs);
    }
    // The single-end phase that infers the paired-end parameters (parallel.cpp:236-262 hasUnambiguousMatchInFirstFile, :276-312
    // processChunkSingleEndForPairInferring, :700-727 and cmb_pair_infer for :329-466): read 1 of every pair in BEST mode; read 2 where read 1 has
    // exactly one match in the first reference file; the pairs whose mates both do are the sample.  The single-end results stay here: the
    // same chunk is then paired from them (samOfChunkPairedBest(..., &inference): pairSingleEndedMatchesBest, searchstrategy.h:1454-1462).
    struct PairedEndInference {
        cmb_pair_inferred inferred{};
        size_t readsGiven = 0, unambiguousPairs = 0;
        struct Single {
            std::vector<cmb_occ> occ;
            std::vector<cmb_aln> aln; // (cigar_off into ops)
            std::vector<uint16_t> ops;
        };
        std::vector<Single> single[2]; // [mate][pair]
        std::vector<uint8_t> read2done;
    };
    template <class Record>
    PairedEndInference inferPairedEndParameters(const std::vector<Record>& mates1, const std::vector<Record>& mates2, uint32_t minIdentity,
                                                uint32_t seqsInFirstFile) {
        if (mates1.size() != mates2.size()) throw std::runtime_error("the two read files do not hold the same number of reads");
        const size_t n = mates1.size();
        PairedEndInference inf;
        inf.single[0].resize(n), inf.single[1].resize(n), inf.read2done.assign(n, 0);
        auto matchSingle = [&](const std::vector<Record>& recs, const std::vector<size_t>& ids, std::vector<typename PairedEndInference::Single>& out) {
            std::string seqs;
            std::vector<uint64_t> offs(ids.size() + 1, 0);
            for (size_t j = 0; j < ids.size(); j++) seqs += recs[ids[j]].read, offs[j + 1] = seqs.size();
            cmb_best* r = nullptr;
            check(cmb_match_best(index.handle(), h, 0, minIdentity, seqs.data(), offs.data(), (uint32_t)ids.size(), &r));
            struct Guard {
                cmb_best* r;
                ~Guard() { cmb_best_destroy(r); }
            } guard{r};
            uint64_t nOcc = 0, nOps = 0;
            check(cmb_best_sizes(r, &nOcc, &nOps));
            std::vector<cmb_occ> occ(nOcc ? nOcc : 1);
            std::vector<cmb_aln> aln(nOcc ? nOcc : 1);
            std::vector<uint16_t> ops(nOps ? nOps : 1);
            std::vector<uint64_t> oo(ids.size() + 1);
            std::vector<uint32_t> best(ids.size() ? ids.size() : 1), hits(ids.size() ? ids.size() : 1);
            check(cmb_best_results(r, occ.data(), aln.data(), occ.size(), ops.data(), ops.size(), oo.data(), best.data(), hits.data(), nullptr));
            for (size_t j = 0; j < ids.size(); j++) {
                typename PairedEndInference::Single& s = out[ids[j]];
                for (uint64_t q = oo[j]; q < oo[j + 1]; q++) {
                    cmb_aln a = aln[q];
                    const uint64_t from = a.cigar_off;
                    a.cigar_off = s.ops.size();
                    a.spans = 0; // (assigned; a trimmed occurrence carries its trimmed coordinates)
                    s.ops.insert(s.ops.end(), ops.begin() + from, ops.begin() + from + a.cigar_len);
                    s.occ.push_back(occ[q]);
                    s.aln.push_back(a);
                }
            }
        };
        // hasUnambiguousMatchInFirstFile: exactly one of the matches lies in the first file; it moves to the front
        auto unambiguous = [&](typename PairedEndInference::Single& s) {
            size_t count = 0, at = 0;
            for (size_t j = 0; j < s.occ.size(); j++)
                if (s.aln[j].seq_id < seqsInFirstFile) {
                    if (++count == 2) return false;
                    at = j;
                }
            if (count == 1) std::swap(s.occ[0], s.occ[at]), std::swap(s.aln[0], s.aln[at]);
            return count == 1;
        };
        std::vector<size_t> all(n);
        for (size_t i = 0; i < n; i++) all[i] = i;
        matchSingle(mates1, all, inf.single[0]);
        std::vector<size_t> second;
        for (size_t i = 0; i < n; i++)
            if (unambiguous(inf.single[0][i])) second.push_back(i), inf.read2done[i] = 1;
        matchSingle(mates2, second, inf.single[1]);
        std::vector<cmb_pair_sample> samples;
        for (size_t i : second)
            if (unambiguous(inf.single[1][i])) {
                const cmb_occ &o1 = inf.single[0][i].occ[0], &o2 = inf.single[1][i].occ[0];
                const cmb_aln &a1 = inf.single[0][i].aln[0], &a2 = inf.single[1][i].aln[0];
                samples.push_back(cmb_pair_sample{a1.seq_begin, a1.seq_begin + (o1.end - o1.begin), o1.strand, a2.seq_begin, a2.seq_begin + (o2.end - o2.begin), o2.strand});
            }
        inf.readsGiven = 2 * n, inf.unambiguousPairs = samples.size();
        check(cmb_pair_infer(samples.data(), samples.size(), &inf.inferred));
        return inf;
    }
    template <class Record>
    std::string samOfChunkPairedBest(const std::vector<Record>& mates1, const std::vector<Record>& mates2, const std::vector<const char*>& seqNames,
                                     uint32_t x, uint32_t minIdentity, uint32_t orientation, uint32_t maxFragSize, uint32_t minFragSize,
                                     bool discordantAllowed, bool unmappedRecords, size_t& mappedPairs, size_t* deviceBatches = nullptr,
                                     const PairedEndInference* startFrom = nullptr) {
        if (mates1.size() != mates2.size()) throw std::runtime_error("the two read files do not hold the same number of reads");
        const uint32_t n = (uint32_t)mates1.size();
        if (startFrom) x = 0; // (pairSingleEndedMatchesBest: no strata beyond the best one)
        uint32_t maxSupported = 0;
        for (; maxSupported < 13; maxSupported++) {
            uint32_t ns = 0, np = 0, crit[16];
            if (cmb_strategy_describe(h, maxSupported + 1, &ns, &np, crit, 16) != CMB_OK || ns == 0) break;
        }
        const std::vector<Record>* in[2] = {&mates1, &mates2};
        std::vector<std::vector<char>> store; // identifiers, reads, reverse complements and reversed qualities as cmb_read_prepare leaves them
        store.reserve((size_t)n * 8);
        std::vector<cmb_pair_read> rd[2];
        for (int m = 0; m < 2; m++)
            for (uint32_t i = 0; i < n; i++) {
                const Record& r = (*in[m])[i];
                const size_t at = store.size();
                store.emplace_back(r.seqID.size() + 1), store.emplace_back(r.read.size() + 1), store.emplace_back(r.read.size() + 1), store.emplace_back(r.qual.size() + 1);
                check(cmb_read_prepare(r.seqID.c_str(), r.read.c_str(), r.qual.c_str(), store[at].data(), store[at + 1].data(), store[at + 2].data(), store[at + 3].data()));
                rd[m].push_back(cmb_pair_read{store[at].data(), store[at + 1].data(), store[at + 2].data(), r.qual.c_str(), store[at + 3].data(), nullptr, 0});
            }
        const cmb_pair_params prm = {orientation, maxFragSize, minFragSize, discordantAllowed ? 1 : 0, unmappedRecords ? 1 : 0};
        cmb_pair_best* pb = nullptr;
        check(cmb_pair_best_create(&prm, x, minIdentity, maxSupported, CMB_METRIC_EDIT, index.handle(), n, rd[0].data(), rd[1].data(), &pb));
        struct Guard {
            cmb_pair_best* p;
            ~Guard() { cmb_pair_best_destroy(p); }
        } guard{pb};
        if (startFrom) {
            if (startFrom->read2done.size() != n) throw std::runtime_error("the single-end results are those of another chunk");
            for (uint32_t i = 0; i < n; i++) {
                const typename PairedEndInference::Single &s1 = startFrom->single[0][i], &s2 = startFrom->single[1][i];
                check(cmb_pair_best_seed(pb, i, s1.occ.data(), s1.aln.data(), s1.occ.size(), s1.ops.data(), s2.occ.data(), s2.aln.data(), s2.occ.size(),
                                         s2.ops.data(), startFrom->read2done[i]));
            }
        }
        std::vector<cmb_pair_request> req(n ? n : 1);
        for (;;) {
            uint64_t nReq = 0;
            check(cmb_pair_best_advance(pb, req.data(), req.size(), &nReq));
            if (nReq == 0) break;
            std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> groups; // (mate, distance) -> pairs
            for (uint64_t j = 0; j < nReq; j++) groups[{req[j].mate, req[j].max_distance}].push_back(req[j].pair);
            for (const auto& g : groups) {
                const uint32_t mate = g.first.first, k = g.first.second;
                std::string seqs;
                std::vector<uint64_t> offs(g.second.size() + 1, 0);
                for (size_t j = 0; j < g.second.size(); j++) seqs += (*in[mate])[g.second[j]].read, offs[j + 1] = seqs.size();
                cmb_batch* b = nullptr;
                check(cmb_batch_create(index.handle(), h, k, seqs.data(), offs.data(), (uint32_t)g.second.size(), &b));
                struct BatchGuard {
                    cmb_batch* b;
                    ~BatchGuard() { cmb_batch_destroy(b); }
                } bg{b};
                check(cmb_batch_want_alignments(b, 1));
                check(cmb_batch_filter_per_strand(b, 1)); // (mapRead filters the strand it searches by itself)
                check(cmb_batch_run(b));
                if (deviceBatches) (*deviceBatches)++;
                uint64_t nOcc = 0, nOps = 0;
                check(cmb_batch_result_size(b, &nOcc));
                std::vector<cmb_occ> occ(nOcc ? nOcc : 1);
                std::vector<cmb_aln> aln(nOcc ? nOcc : 1);
                std::vector<uint64_t> oo(g.second.size() + 1), cnt(CMB_CNT_MAX);
                check(cmb_batch_results(b, occ.data(), occ.size(), oo.data(), cnt.data()));
                (void)cmb_batch_alignments(b, aln.data(), 0, nullptr, 0, &nOps); // sizes first
                std::vector<uint16_t> ops(nOps ? nOps : 1);
                check(cmb_batch_alignments(b, aln.data(), aln.size(), ops.data(), ops.size(), &nOps));
                for (size_t j = 0; j < g.second.size(); j++)
                    for (uint32_t strand = 0; strand < 2; strand++)
                        check(cmb_pair_best_supply(pb, g.second[j], mate, strand, k, occ.data() + oo[j], aln.data() + oo[j], oo[j + 1] - oo[j], ops.data()));
            }
        }
        std::string text;
        std::vector<char> buf;
        for (uint32_t i = 0; i < n; i++) {
            uint32_t nPairs = 0;
            const int64_t len = cmb_pair_best_sam(pb, i, seqNames.data(), nullptr, 0, &nPairs);
            if (len < 0) check((int)len);
            buf.resize((size_t)len + 1);
            cmb_pair_best_sam(pb, i, seqNames.data(), buf.data(), (uint64_t)len + 1, &nPairs);
            text.append(buf.data(), (size_t)len);
            mappedPairs += nPairs > 0;
        }
        return text;
    }
    // the SAM text of a chunk in BEST (+x strata) mode (matchApproxBestPlusX, searchstrategy.cpp:714-746, + generateSE_SAM)
    template <class Record>
    std::string samOfChunkBest(const std::string& seqs, const std::vector<uint64_t>& offs, const std::vector<Record>& recs,
                               const std::vector<std::string>& seqNames, uint32_t x, uint32_t minIdentity, bool unmappedRecords,
                               bool xaTag, size_t& nMapped) {
        const uint32_t nReads = (uint32_t)(offs.size() - 1);
        cmb_best* r = nullptr;
        check(cmb_match_best(index.handle(), h, x, minIdentity, seqs.data(), offs.data(), nReads, &r));
        return samOfBest(r, seqs, offs, recs, seqNames, unmappedRecords, xaTag, nMapped);
    }
    // SearchStrategy::matchApprox (searchstrategy.h:2021-2024), ALL mode, one read
    void matchApprox(ReadBundle& bundle, length_t maxED, Counters& counters, std::vector<TextOcc>& result) {
        std::vector<std::vector<TextOcc>> r;
        matchApproxBatch({bundle}, maxED, counters, r);
        result = r.empty() ? std::vector<TextOcc>() : r[0];
    }
};

struct NamedStrategy : SearchStrategy {
    NamedStrategy(FMIndex& idx, const char* name, PartitionStrategy p, DistanceMetric m) : SearchStrategy(idx) {
        check(cmb_strategy_create_named(name, m, p, &h));
    }
};
struct KucherovKPlus1 : NamedStrategy { // searchstrategy.h:2829
    KucherovKPlus1(FMIndex& idx, PartitionStrategy p, DistanceMetric m) : NamedStrategy(idx, "kuch1", p, m) {}
};
struct PigeonHoleSearchStrategy : NamedStrategy { // searchstrategy.h:3221
    PigeonHoleSearchStrategy(FMIndex& idx, PartitionStrategy p, DistanceMetric m) : NamedStrategy(idx, "pigeon", p, m) {}
};
struct KucherovKPlus2 : NamedStrategy { // searchstrategy.h:2918
    KucherovKPlus2(FMIndex& idx, PartitionStrategy p, DistanceMetric m) : NamedStrategy(idx, "kuch2", p, m) {}
};
struct OptimalKianfar : NamedStrategy { // searchstrategy.h:3026
    OptimalKianfar(FMIndex& idx, PartitionStrategy p, DistanceMetric m) : NamedStrategy(idx, "kianfar", p, m) {}
};
struct O1StarSearchStrategy : NamedStrategy { // searchstrategy.h:3115
    O1StarSearchStrategy(FMIndex& idx, PartitionStrategy p, DistanceMetric m) : NamedStrategy(idx, "01*0", p, m) {}
};
struct MinUSearchStrategy : NamedStrategy { // searchstrategy.h:3284
    MinUSearchStrategy(FMIndex& idx, PartitionStrategy p, DistanceMetric m) : NamedStrategy(idx, "minU", p, m) {}
};
struct DynamicColumbaStrategy : NamedStrategy { // searchstrategy.h:3666 (`-S columba`, the CLI default)
    DynamicColumbaStrategy(FMIndex& idx, PartitionStrategy p, DistanceMetric m) : NamedStrategy(idx, "columba", p, m) {}
};
struct MultipleSchemesStrategy : SearchStrategy { // searchstrategy.h:2584 (`-d <dir>`)
    MultipleSchemesStrategy(FMIndex& idx, const std::string& pathToFolder, PartitionStrategy p, DistanceMetric m)
        : SearchStrategy(idx) {
        check(cmb_strategy_create_from_dir(pathToFolder.c_str(), CMB_DIR_MULTIPLE, m, p, &h));
    }
};
struct CustomSearchStrategy : SearchStrategy { // searchstrategy.h:2130 (`-c <dir> -nD`)
    CustomSearchStrategy(FMIndex& idx, const std::string& pathToFolder, PartitionStrategy p, DistanceMetric m)
        : SearchStrategy(idx) {
        check(cmb_strategy_create_from_dir(pathToFolder.c_str(), CMB_DIR_CUSTOM, m, p, &h));
    }
};
struct DynamicCustomStrategy : SearchStrategy { // searchstrategy.h:3744 (`-c <dir>`)
    DynamicCustomStrategy(FMIndex& idx, const std::string& pathToFolder, PartitionStrategy p, DistanceMetric m)
        : SearchStrategy(idx) {
        check(cmb_strategy_create_from_dir(pathToFolder.c_str(), CMB_DIR_CUSTOM_DYNAMIC, m, p, &h));
    }
};

} // namespace columba_amd

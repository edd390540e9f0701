// columba_amd_bmove.hpp — C++ host adapter over the b-move part of the C-ABI (include/columba_amd.h, section "b-move").
//
// Keeps the names and argument meaning of the reference's run-length compressed index class for the operations that run
// on the device (reference src/, RUN_LENGTH_COMPRESSION flavour, 64-bit length_t):
//
//   reference                                                        here (namespace columba_amd::rlc)
//   ---------------------------------------------------------------  -------------------------------------------------
//   BMove(baseFile, ...) -> fromFiles            bmove/bmove.cpp:45   BMove(baseFile, device)
//   BMove::getCompleteRange                      bmove/bmove.h:369    same
//   BMove::findRangesWithExtraCharForward / Backward /                same names (one parent, one character) and
//          BackwardUniDirectional                bmove.cpp:328-478    extendFMPos for a batch (all four children each)
//   BMove::getTextPositionsFromSARange           bmove.cpp:543-560    same, and a batch variant
//   IndexInterface::exactMatchesOutput           indexinterface.cpp:947   exactMatchesOutput for a chunk of reads
//   SearchStrategy::matchApprox (ALL mode)       searchstrategy.cpp:495   SearchStrategy::matchApproxBatch for a chunk of reads
//     (the RUN_LENGTH_COMPRESSION branches: no in-text verification, locate through the toeholds)
//   SARangePair / MoveRange (SARange)            indexhelpers.h:137-255, :1117   value types with the same accessors
//
// Files: <base>.LFBP and <base>.rev.LFBP are the reference's own (MoveLFReprBP::write).  The reference keeps the samples,
// the phi predecessors and the PLCP in sdsl containers, whose serialisation belongs to sdsl; this loader reads their
// CONTENTS as plain little-endian 64-bit arrays: <base>.smpf.u64 .smpl.u64 .rev.smpf.u64 .rev.smpl.u64 (samples),
// .prdf.u64 .prdl.u64 (marked positions of predFirst / predLast, increasing), .ftr.u64 .ltr.u64, and .plcp.pos.u64 /
// .plcp.sum.u64 (run-length form of the PLCP, see cmb_move_desc).  columba_amd/movebuild.py writes all of them.
//
// Errors are std::runtime_error with the reference's texts where it has them ("Cannot open file: ...").
#pragma once
#include "columba_amd.h"
#include "columba_amd_best.hpp"

#include <cstdint>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace columba_amd {
namespace rlc {

typedef uint64_t length_t;
enum Strand { FORWARD_STRAND = 0, REVERSE_C_STRAND = 1 };

inline void check(int rc) {
    if (rc != CMB_OK) throw std::runtime_error(cmb_last_error());
}

class MoveRange { // indexhelpers.h:137-255
    length_t b, e, bRun, eRun;
    bool valid;

  public:
    MoveRange(length_t begin = 0, length_t end = 0, length_t beginRun = 0, length_t endRun = 0, bool runIndicesValid = true)
        : b(begin), e(end), bRun(beginRun), eRun(endRun), valid(runIndicesValid) {}
    length_t getBegin() const { return b; }
    length_t getEnd() const { return e; }
    length_t getBeginRun() const { return bRun; }
    length_t getEndRun() const { return eRun; }
    bool getRunIndicesValid() const { return valid; }
    bool empty() const { return e <= b; }
    length_t width() const { return empty() ? 0 : e - b; }
    bool operator==(const MoveRange& o) const { return b == o.b && e == o.e && bRun == o.bRun && eRun == o.eRun; }
};
typedef MoveRange SARange;

class SARangePair { // indexhelpers.h:1117-1260 with ToeholdInterface (:1040-1111)
    cmb_move_range r;

  public:
    SARangePair() : r{} { r.runs_valid = r.rev_runs_valid = 1; }
    explicit SARangePair(const cmb_move_range& raw) : r(raw) {}
    SARangePair(const SARange& sa, const SARange& rev, length_t toehold, bool toeholdRepresentsEnd, length_t originalDepth) : r{} {
        r.begin = sa.getBegin(), r.end = sa.getEnd(), r.begin_run = sa.getBeginRun(), r.end_run = sa.getEndRun();
        r.rev_begin = rev.getBegin(), r.rev_end = rev.getEnd(), r.rev_begin_run = rev.getBeginRun(), r.rev_end_run = rev.getEndRun();
        r.runs_valid = sa.getRunIndicesValid(), r.rev_runs_valid = rev.getRunIndicesValid();
        r.toehold = toehold, r.toehold_represents_end = toeholdRepresentsEnd, r.original_depth = (uint32_t)originalDepth;
    }
    SARange getRangeSA() const { return SARange(r.begin, r.end, r.begin_run, r.end_run, r.runs_valid != 0); }
    SARange getRangeSARev() const { return SARange(r.rev_begin, r.rev_end, r.rev_begin_run, r.rev_end_run, r.rev_runs_valid != 0); }
    length_t getToehold() const { return r.toehold; }
    bool getToeholdRepresentsEnd() const { return r.toehold_represents_end != 0; }
    length_t getOriginalDepth() const { return r.original_depth; }
    bool empty() const { return r.end <= r.begin; }
    length_t width() const { return empty() ? 0 : r.end - r.begin; }
    const cmb_move_range& raw() const { return r; }
};

class TextOcc {
    length_t b, e, d;
    Strand s;

  public:
    TextOcc(length_t begin, length_t end, length_t distance, Strand strand) : b(begin), e(end), d(distance), s(strand) {}
    length_t getBegin() const { return b; }
    length_t getEnd() const { return e; }
    length_t getDistance() const { return d; }
    Strand getStrand() const { return s; }
    bool isRevCompl() const { return s == REVERSE_C_STRAND; }
};

class BMove {
    cmb_move_index* h = nullptr;
    length_t textLength = 0, nRuns = 0, nRunsRev = 0;

    static std::vector<uint8_t> slurp(const std::string& fn) {
        std::ifstream f(fn, std::ios::binary);
        if (!f) throw std::runtime_error("Cannot open file: " + fn); // bmove.cpp:53, :62, ...
        return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    }
    static std::vector<uint64_t> slurp64(const std::string& fn) {
        const std::vector<uint8_t> b = slurp(fn);
        std::vector<uint64_t> v(b.size() / 8);
        for (size_t i = 0; i < v.size(); i++) {
            uint64_t x = 0;
            for (int k = 7; k >= 0; k--) x = x << 8 | b[8 * i + k];
            v[i] = x;
        }
        return v;
    }

  public:
    // BMove::fromFiles (bmove.cpp:45-170); lengthBits: width of length_t in the .LFBP files (64 unless built with THIRTY_TWO)
    explicit BMove(const std::string& baseFile, int device = 0, bool withLocate = true, unsigned lengthBits = 64) {
        const std::vector<uint8_t> lf = slurp(baseFile + ".LFBP"), lr = slurp(baseFile + ".rev.LFBP");
        const std::vector<uint64_t> sf = slurp64(baseFile + ".smpf.u64"), sl = slurp64(baseFile + ".smpl.u64"),
                                    rsf = slurp64(baseFile + ".rev.smpf.u64"), rsl = slurp64(baseFile + ".rev.smpl.u64");
        std::vector<uint64_t> pf, ftr, pl, ltr, pp, ps;
        cmb_move_desc d{};
        d.lfbp = lf.data(), d.lfbp_bytes = lf.size(), d.rev_lfbp = lr.data(), d.rev_lfbp_bytes = lr.size(), d.length_bits = lengthBits;
        d.samples_first = sf.data(), d.samples_last = sl.data(), d.rev_samples_first = rsf.data(), d.rev_samples_last = rsl.data();
        if (withLocate) {
            pf = slurp64(baseFile + ".prdf.u64"), ftr = slurp64(baseFile + ".ftr.u64"), pl = slurp64(baseFile + ".prdl.u64");
            ltr = slurp64(baseFile + ".ltr.u64"), pp = slurp64(baseFile + ".plcp.pos.u64"), ps = slurp64(baseFile + ".plcp.sum.u64");
            d.pred_first = pf.data(), d.first_to_run = ftr.data(), d.pred_last = pl.data(), d.last_to_run = ltr.data();
            d.plcp_pos = pp.data(), d.plcp_sum = ps.data(), d.n_plcp = pp.size();
        }
        check(cmb_move_create(&d, device, &h));
        check(cmb_move_info(h, &textLength, &nRuns, &nRunsRev));
        if (sf.size() != nRuns || sl.size() != nRuns || rsf.size() != nRunsRev || rsl.size() != nRunsRev) {
            cmb_move_destroy(h);
            throw std::runtime_error("sample arrays do not have one entry per run");
        }
    }
    ~BMove() { cmb_move_destroy(h); }
    BMove(const BMove&) = delete;
    BMove& operator=(const BMove&) = delete;

    length_t getTextSize() const { return textLength; }
    length_t getNumberOfRuns() const { return nRuns; }
    length_t getSwitchPoint() const { return 0; } // bmove.cpp:172-174: no in-text verification with b-move
    cmb_move_index* handle() const { return h; }
    // The text beside the index: what the CIGARs and SAM records of the occurrences are computed on (the reference's search carries
    // the matched string instead, indexinterface.h:294-303; here 288 GB of HBM hold a 2-bit copy of the text next to the tables).
    // seqStarts: begin of every sequence in the concatenated text + the final n - 1 (the .pos file, indexinterface.cpp:162), or empty.
    void attachText(const std::string& text, const std::vector<uint32_t>& seqStarts = {}) {
        check(cmb_move_attach_text(h, text.data(), text.size(), seqStarts.empty() ? nullptr : seqStarts.data(), (uint32_t)seqStarts.size()));
    }

    SARangePair getCompleteRange() const { // bmove.h:369-373
        cmb_move_range r;
        check(cmb_move_complete_range(h, &r));
        return SARangePair(r);
    }

    // IndexInterface::extendFMPos (indexinterface.cpp:675-697) for a batch: children[4 i + c - 1] is parent i extended with
    // character c (1..4 = ACGT); mode 0 forward, 1 backward, 2 uni-directional backward
    void extendFMPos(const std::vector<SARangePair>& parents, int mode, std::vector<SARangePair>& children, std::vector<uint8_t>& ok) const {
        std::vector<cmb_move_range> in(parents.size()), out(4 * parents.size());
        for (size_t i = 0; i < parents.size(); i++) in[i] = parents[i].raw();
        ok.assign(4 * parents.size(), 0);
        check(cmb_move_extend_batch(h, mode, in.data(), in.size(), out.data(), ok.data()));
        children.clear();
        for (const auto& r : out) children.emplace_back(r);
    }
    bool extendOne(int mode, length_t positionInAlphabet, const SARangePair& parent, SARangePair& child) const {
        if (positionInAlphabet < 1 || positionInAlphabet > 4) throw std::runtime_error("character outside the alphabet");
        std::vector<SARangePair> ch;
        std::vector<uint8_t> ok;
        extendFMPos({parent}, mode, ch, ok);
        child = ch[positionInAlphabet - 1];
        return ok[positionInAlphabet - 1] != 0;
    }
    bool findRangesWithExtraCharForward(length_t c, const SARangePair& p, SARangePair& ch) const { return extendOne(0, c, p, ch); }   // bmove.cpp:384
    bool findRangesWithExtraCharBackward(length_t c, const SARangePair& p, SARangePair& ch) const { return extendOne(1, c, p, ch); }  // bmove.cpp:328
    bool findRangesWithExtraCharBackwardUniDirectional(length_t c, const SARangePair& p, SARangePair& ch) const { return extendOne(2, c, p, ch); } // :444

    // bmove.cpp:543-560; the batch variant returns the positions of range i in positions[i]
    void getTextPositionsFromSARange(const SARangePair& ranges, std::vector<length_t>& positions) const {
        std::vector<std::vector<length_t>> all;
        getTextPositionsFromSARanges({ranges}, all);
        positions = all[0];
    }
    void getTextPositionsFromSARanges(const std::vector<SARangePair>& ranges, std::vector<std::vector<length_t>>& positions) const {
        std::vector<cmb_move_range> in(ranges.size());
        std::vector<uint64_t> off(ranges.size() + 1, 0);
        for (size_t i = 0; i < ranges.size(); i++) in[i] = ranges[i].raw(), off[i + 1] = off[i] + ranges[i].width();
        std::vector<uint64_t> flat(off.back());
        check(cmb_move_locate_batch(h, in.data(), in.size(), off.data(), flat.data()));
        positions.assign(ranges.size(), {});
        for (size_t i = 0; i < ranges.size(); i++) positions[i].assign(flat.begin() + off[i], flat.begin() + off[i + 1]);
    }

    // indexinterface.cpp:947-1014 for a chunk of reads, both strands (searchstrategy.cpp:499-510): matches[i] = the occurrences
    // of read i, forward strand first; returns NODE_COUNTER
    uint64_t exactMatchesOutput(const std::vector<std::string>& reads, std::vector<std::vector<TextOcc>>& matches) const {
        std::string buf;
        std::vector<uint64_t> off(reads.size() + 1, 0), occOff(reads.size() + 1, 0);
        for (size_t i = 0; i < reads.size(); i++) buf += reads[i], off[i + 1] = buf.size();
        std::vector<cmb_move_occ> occ(reads.size() * 2 + 64);
        uint64_t nOcc = 0, counters[2] = {0, 0};
        int rc = cmb_move_match_exact(h, buf.data(), off.data(), reads.size(), occ.data(), occ.size(), occOff.data(), &nOcc, counters);
        if (rc == CMB_ERR_OVERFLOW) {
            occ.resize(nOcc);
            rc = cmb_move_match_exact(h, buf.data(), off.data(), reads.size(), occ.data(), occ.size(), occOff.data(), &nOcc, counters);
        }
        check(rc);
        matches.assign(reads.size(), {});
        for (size_t i = 0; i < reads.size(); i++)
            for (uint64_t j = occOff[i]; j < occOff[i + 1]; j++)
                matches[i].emplace_back(occ[j].begin, occ[j].end, occ[j].distance, occ[j].strand ? REVERSE_C_STRAND : FORWARD_STRAND);
        return counters[0];
    }
};

// SearchStrategy of the RLC flavour for the path that runs on the device: the body of processChunk's loop (parallel.cpp:67-78)
// for a whole chunk in ALL mode.  name: one of the reference's -S names ("columba", "multiple_opt", "kuch1", "kuch2", "kianfar",
// "01*0", "pigeon", "minU"; alignparameters.cpp:1341-1372); partitioning 0 uniform, 1 static, 2 dynamic; metric 0 Hamming, 1 edit.
class SearchStrategy {
    BMove& index;
    cmb_strategy* h = nullptr;
    unsigned kmerSize;

  public:
    SearchStrategy(BMove& idx, const std::string& name, int partitioning = CMB_PARTITION_DYNAMIC, int metric = CMB_METRIC_EDIT,
                   unsigned kmerSize_ = 10)
        : index(idx), kmerSize(kmerSize_) {
        check(cmb_strategy_create_named(name.c_str(), metric, partitioning, &h));
    }
    SearchStrategy(const SearchStrategy&) = delete;
    ~SearchStrategy() { cmb_strategy_destroy(h); }
    // The SAM records of a chunk in ALL mode (generateOutputSingleEnd, searchstrategy.cpp:1824-1902): occurrences, their CIGARs
    // (findCIGAR on text[begin, end) — the matched string of this flavour, indexinterface.h:294-303) and sequence names.  Needs
    // BMove::attachText.  ids / quals as read from the FASTQ file; seqNames: the sequences of the index in text order.
    std::string samOfChunk(const std::vector<std::string>& ids, const std::vector<std::string>& reads, const std::vector<std::string>& quals,
                           const std::vector<std::string>& seqNames, length_t maxED, int metric = CMB_METRIC_EDIT, bool unmappedRecords = true,
                           bool xaTag = false) {
        std::string buf;
        std::vector<uint64_t> off(reads.size() + 1, 0), occOff(reads.size() + 1, 0);
        for (size_t i = 0; i < reads.size(); i++) buf += reads[i], off[i + 1] = buf.size();
        cmb_move_batch* b = nullptr;
        check(cmb_move_batch_create(index.handle(), h, (uint32_t)maxED, kmerSize, buf.data(), off.data(), (uint32_t)reads.size(), &b));
        struct Guard {
            cmb_move_batch* b;
            ~Guard() { cmb_move_batch_destroy(b); }
        } guard{b};
        check(cmb_move_batch_want_alignments(b, 1));
        check(cmb_move_batch_run(b));
        uint64_t n = 0, nOps = 0;
        check(cmb_move_batch_result_size(b, &n));
        std::vector<cmb_move_occ> occ(n ? n : 1);
        check(cmb_move_batch_results(b, occ.data(), occ.size(), occOff.data(), nullptr));
        std::vector<cmb_aln> aln(n ? n : 1);
        (void)cmb_move_batch_alignments(b, aln.data(), 0, nullptr, 0, &nOps); // (sizes first)
        std::vector<uint16_t> ops(nOps ? nOps : 1);
        check(cmb_move_batch_alignments(b, aln.data(), aln.size(), ops.data(), ops.size(), &nOps));
        std::vector<cmb_occ> occ32(n ? n : 1);
        for (uint64_t i = 0; i < n; i++) occ32[i] = cmb_occ{(uint32_t)occ[i].begin, (uint32_t)occ[i].end, occ[i].distance, occ[i].strand};
        std::vector<const char*> pi, pq, pn;
        for (const auto& x : ids) pi.push_back(x.c_str());
        for (const auto& x : quals) pq.push_back(x.c_str());
        for (const auto& x : seqNames) pn.push_back(x.c_str());
        cmb_index* tix = cmb_move_text_index(index.handle());
        const int64_t len = cmb_sam_chunk(tix, (uint32_t)maxED, metric, buf.data(), off.data(), (uint32_t)reads.size(), pi.data(), pq.data(), pn.data(),
                                          occ32.data(), occOff.data(), aln.data(), ops.data(), unmappedRecords, xaTag, nullptr, 0);
        if (len < 0) check((int)len);
        std::string out((size_t)len + 1, '\0');
        (void)cmb_sam_chunk(tix, (uint32_t)maxED, metric, buf.data(), off.data(), (uint32_t)reads.size(), pi.data(), pq.data(), pn.data(), occ32.data(),
                            occOff.data(), aln.data(), ops.data(), unmappedRecords, xaTag, &out[0], out.size());
        out.resize((size_t)len);
        return out;
    }
    // The SAM records of a chunk in BEST (+x strata) mode — the reference's default, `-a best` (matchApproxBestPlusX,
    // searchstrategy.cpp:714-746, + generateSE_SAM): cmb_move_match_best.  Needs BMove::attachText.
    struct BestRecord {
        std::string seqID, qual;
    };
    std::string samOfChunkBest(const std::vector<std::string>& ids, const std::vector<std::string>& reads, const std::vector<std::string>& quals,
                               const std::vector<std::string>& seqNames, uint32_t x, uint32_t minIdentity, bool unmappedRecords, bool xaTag,
                               size_t& nMapped) {
        std::string buf;
        std::vector<uint64_t> off(reads.size() + 1, 0);
        std::vector<BestRecord> recs(reads.size());
        for (size_t i = 0; i < reads.size(); i++) buf += reads[i], off[i + 1] = buf.size(), recs[i] = BestRecord{ids[i], quals[i]};
        cmb_best* r = nullptr;
        check(cmb_move_match_best(index.handle(), h, x, minIdentity, kmerSize, buf.data(), off.data(), (uint32_t)reads.size(), &r));
        return samOfBest(r, buf, off, recs, seqNames, unmappedRecords, xaTag, nMapped);
    }
    // Read pairs in BEST (+x strata) mode on this flavour (matchApproxPairedEndBestPlusX, searchstrategy.cpp:1091-1179): the pairs walk through
    // their strata together (cmb_pair_best_*); every round the lists the unfinished pairs wait for come from one b-move batch per (mate,
    // distance) over the reads that ask — ALL mode, every strand filtered by itself, with alignments.  Needs BMove::attachText.
    // orientation: CMB_ORIENTATION_*.
    std::string samOfChunkPairedBest(const std::vector<std::string>& ids1, const std::vector<std::string>& reads1, const std::vector<std::string>& quals1,
                                     const std::vector<std::string>& ids2, const std::vector<std::string>& reads2, const std::vector<std::string>& quals2,
                                     const std::vector<std::string>& seqNames, uint32_t x, uint32_t minIdentity, uint32_t orientation,
                                     uint32_t maxFragSize, uint32_t minFragSize, bool discordantAllowed, bool unmappedRecords, size_t& mappedPairs) {
        if (reads1.size() != reads2.size()) throw std::runtime_error("the two read files do not hold the same number of reads");
        const uint32_t n = (uint32_t)reads1.size();
        uint32_t maxSupported = 0;
        for (; maxSupported < 13; maxSupported++) {
            uint32_t ns = 0, np = 0, crit[16];
            if (cmb_strategy_describe(h, maxSupported + 1, &ns, &np, crit, 16) != CMB_OK || ns == 0) break;
        }
        const std::vector<std::string>* R[2] = {&reads1, &reads2};
        const std::vector<std::string>* I[2] = {&ids1, &ids2};
        const std::vector<std::string>* Q[2] = {&quals1, &quals2};
        std::vector<std::vector<char>> store;
        std::vector<cmb_pair_read> rd[2];
        for (int m = 0; m < 2; m++)
            for (uint32_t i = 0; i < n; i++) {
                const std::string &id = (*I[m])[i], &seq = (*R[m])[i], &q = (*Q[m])[i];
                const size_t at = store.size();
                store.emplace_back(id.size() + 1), store.emplace_back(seq.size() + 1), store.emplace_back(seq.size() + 1), store.emplace_back(q.size() + 1);
                check(cmb_read_prepare(id.c_str(), seq.c_str(), q.c_str(), store[at].data(), store[at + 1].data(), store[at + 2].data(), store[at + 3].data()));
                rd[m].push_back(cmb_pair_read{store[at].data(), store[at + 1].data(), store[at + 2].data(), q.c_str(), store[at + 3].data(), nullptr, 0});
            }
        const cmb_pair_params prm = {orientation, maxFragSize, minFragSize, discordantAllowed ? 1 : 0, unmappedRecords ? 1 : 0};
        cmb_pair_best* pb = nullptr;
        check(cmb_pair_best_create(&prm, x, minIdentity, maxSupported, CMB_METRIC_EDIT, cmb_move_text_index(index.handle()), n, rd[0].data(), rd[1].data(), &pb));
        struct Guard {
            cmb_pair_best* p;
            ~Guard() { cmb_pair_best_destroy(p); }
        } guard{pb};
        std::vector<cmb_pair_request> req(n ? n : 1);
        for (;;) {
            uint64_t nReq = 0;
            check(cmb_pair_best_advance(pb, req.data(), req.size(), &nReq));
            if (nReq == 0) break;
            std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> groups; // (mate, distance) -> pairs
            for (uint64_t j = 0; j < nReq; j++) groups[{req[j].mate, req[j].max_distance}].push_back(req[j].pair);
            for (const auto& g : groups) {
                const uint32_t mate = g.first.first, k = g.first.second;
                std::string buf;
                std::vector<uint64_t> off(g.second.size() + 1, 0), occOff(g.second.size() + 1, 0);
                for (size_t j = 0; j < g.second.size(); j++) buf += (*R[mate])[g.second[j]], off[j + 1] = buf.size();
                cmb_move_batch* b = nullptr;
                check(cmb_move_batch_create(index.handle(), h, k, kmerSize, buf.data(), off.data(), (uint32_t)g.second.size(), &b));
                struct BatchGuard {
                    cmb_move_batch* b;
                    ~BatchGuard() { cmb_move_batch_destroy(b); }
                } bg{b};
                check(cmb_move_batch_want_alignments(b, 1));
                check(cmb_move_batch_filter_per_strand(b, 1));
                check(cmb_move_batch_run(b));
                uint64_t nOcc = 0, nOps = 0;
                check(cmb_move_batch_result_size(b, &nOcc));
                std::vector<cmb_move_occ> occ(nOcc ? nOcc : 1);
                check(cmb_move_batch_results(b, occ.data(), occ.size(), occOff.data(), nullptr));
                std::vector<cmb_aln> aln(nOcc ? nOcc : 1);
                (void)cmb_move_batch_alignments(b, aln.data(), 0, nullptr, 0, &nOps); // (sizes first)
                std::vector<uint16_t> ops(nOps ? nOps : 1);
                check(cmb_move_batch_alignments(b, aln.data(), aln.size(), ops.data(), ops.size(), &nOps));
                std::vector<cmb_occ> occ32(nOcc ? nOcc : 1);
                for (uint64_t i = 0; i < nOcc; i++) occ32[i] = cmb_occ{(uint32_t)occ[i].begin, (uint32_t)occ[i].end, occ[i].distance, occ[i].strand};
                for (size_t j = 0; j < g.second.size(); j++)
                    for (uint32_t strand = 0; strand < 2; strand++)
                        check(cmb_pair_best_supply(pb, g.second[j], mate, strand, k, occ32.data() + occOff[j], aln.data() + occOff[j], occOff[j + 1] - occOff[j],
                                                   ops.data()));
            }
        }
        std::vector<const char*> pn;
        for (const auto& name : seqNames) pn.push_back(name.c_str());
        std::string text;
        std::vector<char> out;
        for (uint32_t i = 0; i < n; i++) {
            uint32_t nPairs = 0;
            const int64_t len = cmb_pair_best_sam(pb, i, pn.data(), nullptr, 0, &nPairs);
            if (len < 0) check((int)len);
            out.resize((size_t)len + 1);
            cmb_pair_best_sam(pb, i, pn.data(), out.data(), (uint64_t)len + 1, &nPairs);
            text.append(out.data(), (size_t)len);
            mappedPairs += nPairs > 0;
        }
        return text;
    }
    // matches[i] = the occurrences of reads[i] as filterPtr leaves them (searchstrategy.cpp:529); counters[CMB_CNT_*]
    void matchApproxBatch(const std::vector<std::string>& reads, length_t maxED, std::vector<uint64_t>& counters,
                          std::vector<std::vector<TextOcc>>& matches) {
        std::string buf;
        std::vector<uint64_t> off(reads.size() + 1, 0), occOff(reads.size() + 1, 0);
        for (size_t i = 0; i < reads.size(); i++) buf += reads[i], off[i + 1] = buf.size();
        std::vector<cmb_move_occ> occ(reads.size() * 4 + 64);
        counters.assign(CMB_CNT_MAX, 0);
        uint64_t needed = 0;
        int rc = cmb_move_match_batch(index.handle(), h, (uint32_t)maxED, kmerSize, buf.data(), off.data(), (uint32_t)reads.size(), occ.data(),
                                      occ.size(), occOff.data(), counters.data(), &needed);
        if (rc == CMB_ERR_OVERFLOW) {
            occ.resize(needed);
            rc = cmb_move_match_batch(index.handle(), h, (uint32_t)maxED, kmerSize, buf.data(), off.data(), (uint32_t)reads.size(), occ.data(),
                                      occ.size(), occOff.data(), counters.data(), &needed);
        }
        check(rc);
        matches.assign(reads.size(), {});
        for (size_t i = 0; i < reads.size(); i++)
            for (uint64_t j = occOff[i]; j < occOff[i + 1]; j++)
                matches[i].emplace_back(occ[j].begin, occ[j].end, occ[j].distance, occ[j].strand ? REVERSE_C_STRAND : FORWARD_STRAND);
    }
};

} // namespace rlc
} // namespace columba_amd

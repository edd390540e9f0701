// SAM records of a chunk from a BEST (+x strata) result (cmb_match_best on the FM-index, cmb_move_match_best on the b-move index):
// generateOutputSingleEnd + generateSE_SAM / generateSE_SAM_XATag of the reference (src/searchstrategy.cpp:1824-1902,
// src/searchstrategy.h:1612-1641) through the record builders of the C-ABI.  Shared by include/columba_amd.hpp and
// include/columba_amd_bmove.hpp; takes ownership of the result.
#pragma once
#include "columba_amd.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace columba_amd {

template <class Record>
std::string samOfBest(cmb_best* r, const std::string& seqs, const std::vector<uint64_t>& offs, const std::vector<Record>& recs,
                      const std::vector<std::string>& seqNames, bool unmappedRecords, bool xaTag, size_t& nMapped) {
    auto check = [](int rc) {
        if (rc != CMB_OK) throw std::runtime_error(cmb_last_error());
    };
    const uint32_t nReads = (uint32_t)(offs.size() - 1);
    struct Guard {
        cmb_best* r;
        ~Guard() { cmb_best_destroy(r); }
    } guard{r};
    uint64_t nOcc = 0, nOps = 0;
    check(cmb_best_sizes(r, &nOcc, &nOps));
    std::vector<cmb_occ> occ(nOcc ? nOcc : 1);
    std::vector<cmb_aln> aln(nOcc ? nOcc : 1);
    std::vector<uint16_t> ops(nOps ? nOps : 1);
    std::vector<uint64_t> oo(nReads + 1);
    std::vector<uint32_t> best(nReads ? nReads : 1), hits(nReads ? nReads : 1);
    check(cmb_best_results(r, occ.data(), aln.data(), occ.size(), ops.data(), ops.size(), oo.data(), best.data(), hits.data(),
                           nullptr));
    std::string text;
    std::vector<char> buf;
    auto emit = [&](int64_t n, auto&& call) {
        if (n < 0) check((int)n);
        buf.resize((size_t)n + 1);
        call(buf.data(), (uint64_t)n + 1);
        text.append(buf.data(), (size_t)n);
    };
    for (uint32_t i = 0; i < nReads; i++) {
        const size_t len = (size_t)(offs[i + 1] - offs[i]);
        std::vector<char> id(recs[i].seqID.size() + 2), rd(len + 2), rc(len + 2), rq(recs[i].qual.size() + 2);
        const std::string raw(seqs.data() + offs[i], len);
        check(cmb_read_prepare(recs[i].seqID.c_str(), raw.c_str(), recs[i].qual.c_str(), id.data(), rd.data(), rc.data(), rq.data()));
        if (oo[i + 1] == oo[i]) {
            if (unmappedRecords)
                emit(cmb_sam_unmapped_se(id.data(), rd.data(), recs[i].qual.c_str(), nullptr, 0),
                     [&](char* o, uint64_t c) { cmb_sam_unmapped_se(id.data(), rd.data(), recs[i].qual.c_str(), o, c); });
            continue;
        }
        nMapped++;
        std::vector<cmb_sam_hit> hs;
        for (uint64_t j = oo[i]; j < oo[i + 1]; j++)
            hs.push_back(cmb_sam_hit{seqNames[aln[j].seq_id].c_str(), aln[j].seq_begin, occ[j].distance, occ[j].strand,
                                     ops.data() + aln[j].cigar_off, aln[j].cigar_len});
        const bool rcFirst = occ[oo[i]].strand != 0;
        const char* ps = rcFirst ? rc.data() : rd.data();
        const char* pq = rcFirst ? rq.data() : recs[i].qual.c_str();
        if (xaTag) {
            emit(cmb_sam_se_xa(id.data(), hs.data(), (uint32_t)hs.size(), hits[i], ps, pq, nullptr, 0),
                 [&](char* o, uint64_t c) { cmb_sam_se_xa(id.data(), hs.data(), (uint32_t)hs.size(), hits[i], ps, pq, o, c); });
        } else { // generateSE_SAM (searchstrategy.h:1633-1641): the first record primary, the others secondary
            emit(cmb_sam_se(id.data(), &hs[0], 1, hits[i], best[i], ps, pq, nullptr, 0),
                 [&](char* o, uint64_t c) { cmb_sam_se(id.data(), &hs[0], 1, hits[i], best[i], ps, pq, o, c); });
            for (size_t j = 1; j < hs.size(); j++)
                emit(cmb_sam_se(id.data(), &hs[j], 0, hits[i], best[i], "*", "*", nullptr, 0),
                     [&](char* o, uint64_t c) { cmb_sam_se(id.data(), &hs[j], 0, hits[i], best[i], "*", "*", o, c); });
        }
    }
    return text;
}

} // namespace columba_amd

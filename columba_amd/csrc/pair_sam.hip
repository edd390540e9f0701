// SAM records of paired-end reads (include/columba_amd.h, section "paired-end records"): host code, a translation unit of its own.
// Reference: TextOcc::generateSAMPairedEnd (indexhelpers.cpp:114-166), createUnmappedSAMOccurrencePE (:186-213),
// generateSAMUnpaired (:215-262), getFlagsPE (indexhelpers.h:340-371), getMapQ / getMapQPairedEnd (:378-410).
#include "../../include/columba_amd.h"
#include "host_sam.hpp"

#include <cstring>
#include <string>

namespace cmb {
int failWith(int code, const std::string& msg); // columba_amd.hip
}
using namespace cmb;

namespace {
int64_t putText(const std::string& s, char* out, uint64_t cap) { // length of the text; written (with its NUL) if it fits
    if (out && cap > s.size()) std::memcpy(out, s.c_str(), s.size() + 1);
    return (int64_t)s.size();
}
SamHit hitOf(const cmb_sam_hit& h) {
    SamHit r;
    r.seqName = h.seq_name ? h.seq_name : "*";
    r.cigar = cigarString(h.cigar_ops, h.n_ops);
    r.pos0 = h.pos0;
    r.distance = h.distance;
    r.revCompl = h.revcomp != 0;
    return r;
}
} // namespace

extern "C" int64_t cmb_sam_pe(const char* read_id, const cmb_sam_hit* hit, int first_in_pair, const cmb_sam_hit* mate, uint32_t n_pairs,
                              uint32_t min_score, uint32_t frag_size, int discordant, int primary, const char* print_seq,
                              const char* print_qual, char* out, uint64_t cap) {
    if (!read_id || !hit || !print_seq || !print_qual || n_pairs == 0) return failWith(CMB_ERR_INVALID, "bad argument");
    const SamHit h = hitOf(*hit);
    const bool mateMapped = mate != nullptr;
    SamHit m;
    if (mateMapped) m = hitOf(*mate);
    // getFlagsPE: an unmapped mate lies on the forward strand and has distance 0 (indexhelpers.cpp:190-193)
    unsigned flags = 1u;
    flags |= (!discordant && mateMapped ? 1u : 0u) << 1;
    flags |= (mateMapped ? 0u : 1u) << 3;
    flags |= (h.revCompl ? 1u : 0u) << 4;
    flags |= (mateMapped && m.revCompl ? 1u : 0u) << 5;
    flags |= (first_in_pair ? 1u : 0u) << 6;
    flags |= (first_in_pair ? 0u : 1u) << 7; // the mate is the other read of the pair
    flags |= (primary ? 0u : 1u) << 8;
    int mapq = 0; // getMapQPairedEnd
    if (!(h.distance + (mateMapped ? m.distance : 0u) > min_score)) mapq = n_pairs == 1 ? MAX_MAPQ : (int)std::round(-10.0 * std::log10(1 - 1.0 / n_pairs));
    std::string qual = print_qual;
    if (qual.empty()) qual = "*";
    std::string o;
    o.reserve(std::strlen(read_id) + std::strlen(print_seq) + qual.size() + 150);
    o += read_id;
    o += '\t';
    o += std::to_string(flags);
    o += '\t';
    o += h.seqName;
    o += '\t';
    o += std::to_string(h.pos0 + 1);
    o += '\t';
    o += std::to_string(mapq);
    o += '\t';
    o += h.cigar;
    o += '\t';
    o += mateMapped ? m.seqName : std::string("*");
    o += '\t';
    o += std::to_string(mateMapped ? m.pos0 + 1 : 0u);
    o += '\t';
    if (mateMapped && h.pos0 > m.pos0) o += '-';
    o += std::to_string(mateMapped ? frag_size : 0u);
    o += '\t';
    o += print_seq;
    o += '\t';
    o += qual;
    o += "\tAS:i:";
    o += std::to_string(h.distance);
    o += "\tNM:i:";
    o += std::to_string(h.distance);
    o += "\tPG:Z:Columba\n";
    return putText(o, out, cap);
}

extern "C" int64_t cmb_sam_unpaired(const char* read_id, const cmb_sam_hit* hit, int first_in_pair, uint32_t n_hits, uint32_t min_score,
                                    int primary, const char* print_seq, const char* print_qual, char* out, uint64_t cap) {
    if (!read_id || !hit || !print_seq || !print_qual) return failWith(CMB_ERR_INVALID, "bad argument");
    const SamHit h = hitOf(*hit);
    // (no strand flag on these records, and the sequence only on the primary line: indexhelpers.cpp:224-239)
    const unsigned flags = (1u + (first_in_pair ? 64u : 128u)) | (primary ? 0u : 256u);
    std::string seq = primary ? print_seq : "*", qual = primary ? print_qual : "*";
    if (qual.empty()) qual = "*";
    std::string o = std::string(read_id) + "\t" + std::to_string(flags) + "\t" + h.seqName + "\t" + std::to_string(h.pos0 + 1) + "\t" +
                    std::to_string(mapQ(h.distance, n_hits, min_score)) + "\t" + h.cigar + "\t*\t0\t0\t" + seq + "\t" + qual + "\tAS:i:" +
                    std::to_string(h.distance) + "\tNM:i:" + std::to_string(h.distance) + "\tPG:Z:Columba\n";
    return putText(o, out, cap);
}

extern "C" int64_t cmb_sam_unmapped_pe(const char* read_id, const char* seq, const char* qual, int first_in_pair, int mate_mapped,
                                       int mate_revcomp, char* out, uint64_t cap) {
    if (!read_id || !seq || !qual) return failWith(CMB_ERR_INVALID, "bad argument");
    const unsigned flags = 1u | 4u | (mate_mapped ? 0u : 8u) | (mate_revcomp ? 32u : 0u) | (first_in_pair ? 64u : 128u);
    return putText(std::string(read_id) + "\t" + std::to_string(flags) + "\t*\t0\t0\t*\t*\t0\t0\t" + seq + "\t" + qual + "\tPG:Z:Columba\n", out, cap);
}

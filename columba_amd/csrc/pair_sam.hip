// SAM records of paired-end reads (include/columba_amd.h, section "paired-end records"): host code, a translation unit of its own.
// Reference: TextOcc::generateSAMPairedEnd (indexhelpers.cpp:114-166), createUnmappedSAMOccurrencePE (:186-213),
// generateSAMUnpaired (:215-262), getFlagsPE (indexhelpers.h:340-371), getMapQ / getMapQPairedEnd (:378-410).
#include "../../include/columba_amd.h"
#include "host_sam.hpp"

#include <cmath>
#include <cstring>
#include <functional>
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

// ------------------------------------------------------------------------------------------------------------------------------
// Pairing of the single-end occurrences of two mates in ALL mode: SearchStrategy::pairSingleEndedMatchesAll
// (searchstrategy.cpp:1345-1399) -> processComb{FR,RF,FF}All (searchstrategy.h:753-861) -> pairOccurrences (:1281-1344);
// without a concordant pair pairDiscordantly (:1586-1646) -> addDiscPairs (:1518-1585) / addUnpairedMatches (:1401-1462,
// searchstrategy.h:1178-1230) / addOneUnmapped (:1463-1517) / addBothUnmapped (searchstrategy.h:1236-1247); the records by
// generateSAMPairedEnd (:1904-1970) in the order OutputWriter::writeChunks prints them (fastq.cpp:662-702).
// The occurrences arrive sequence-assigned (cmb_batch_alignments): the reference assigns lazily, candidate by candidate
// (assignSequenceAndCIGAR), with the same outcome per occurrence.
// ------------------------------------------------------------------------------------------------------------------------------
#include <algorithm>
#include <vector>

namespace {
struct POcc {
    const cmb_pair_occ* o;
    bool second; // read 2 of the pair
    bool assigned() const { return o->seq_id != 0xFFFFFFFFu; }
    uint32_t width() const { return o->end - o->begin; }
    uint32_t indexEnd() const { return o->index_begin + width(); }
};
struct PPair {
    POcc up, down; // down.o == nullptr: not mapped (its record is the unmapped one)
    bool upValid, downValid;
    uint32_t fragSize, distance;
    bool discordant;
    std::string upLine, downLine;
};
bool occLess(const POcc& a, const POcc& b) { // TextOcc::operator< (indexhelpers.h:776-792) on the occurrences as located
    if (a.o->index_begin != b.o->index_begin) return a.o->index_begin < b.o->index_begin;
    if (a.o->distance != b.o->distance) return a.o->distance < b.o->distance;
    return a.width() < b.width();
}
cmb_sam_hit hitFrom(const POcc& p, const char* const* seqNames) {
    cmb_sam_hit h;
    h.seq_name = seqNames[p.o->seq_id];
    h.pos0 = p.o->begin;
    h.distance = p.o->distance;
    h.revcomp = p.o->strand;
    h.cigar_ops = p.o->cigar_ops;
    h.n_ops = p.o->n_ops;
    return h;
}
std::string callText(const std::function<int64_t(char*, uint64_t)>& f) {
    std::string s((size_t)f(nullptr, 0), '\0');
    std::vector<char> buf(s.size() + 1);
    f(buf.data(), buf.size());
    return std::string(buf.data(), s.size());
}
} // namespace

extern "C" int64_t cmb_pair_sam(const cmb_pair_params* prm, const cmb_pair_read* r1, const cmb_pair_read* r2, const char* const* seq_names,
                                char* out, uint64_t cap, uint32_t* n_pairs_out) {
    if (!prm || !r1 || !r2 || !seq_names || prm->orientation > 2 || !r1->id || !r1->seq || !r1->revcomp || !r2->id || !r2->seq || !r2->revcomp ||
        (r1->n_occ && !r1->occ) || (r2->n_occ && !r2->occ))
        return failWith(CMB_ERR_INVALID, "bad argument");
    const cmb_pair_read* R[2] = {r1, r2};
    std::vector<POcc> st[2][2]; // [read][strand]
    for (int r = 0; r < 2; r++)
        for (uint32_t i = 0; i < R[r]->n_occ; i++) {
            const cmb_pair_occ& o = R[r]->occ[i];
            if (o.end < o.begin || o.strand > 1) return failWith(CMB_ERR_INVALID, "bad occurrence");
            st[r][o.strand].push_back(POcc{&o, r == 1});
        }
    for (int r = 0; r < 2; r++)
        for (int s = 0; s < 2; s++) std::stable_sort(st[r][s].begin(), st[r][s].end(), occLess); // searchstrategy.cpp:1368-1378 (its DEVELOPER_MODE order: ties keep the caller's order)
    std::vector<PPair> pairs;
    // pairOccurrences (searchstrategy.cpp:1281-1344)
    auto pairOccurrences = [&](const std::vector<POcc>& U, const std::vector<POcc>& D) {
        if (U.empty() || D.empty()) return;
        for (const POcc& u : U) {
            const uint32_t upos = u.o->index_begin;
            auto it = std::lower_bound(D.begin(), D.end(), upos, [](const POcc& d, uint32_t p) { return d.o->index_begin < p; });
            for (; it != D.end(); ++it) {
                const uint32_t frag = it->indexEnd() - upos;
                if (frag <= prm->max_frag && frag >= prm->min_frag) {
                    if (!u.assigned()) break;
                    if (!it->assigned()) continue;
                    if (u.o->seq_id != it->o->seq_id) continue;
                    pairs.push_back(PPair{u, *it, true, true, it->o->end - u.o->begin, u.o->distance + it->o->distance, false, "", ""});
                } else if (frag > prm->max_frag)
                    break;
            }
        }
    };
    std::vector<POcc>&fw1 = st[0][0], &rc1 = st[0][1], &fw2 = st[1][0], &rc2 = st[1][1];
    if (prm->orientation == CMB_ORIENTATION_FR) { // searchstrategy.h:790-803
        pairOccurrences(fw1, rc2);
        pairOccurrences(fw2, rc1);
    } else if (prm->orientation == CMB_ORIENTATION_FF) { // :819-832
        pairOccurrences(fw1, fw2);
        pairOccurrences(rc2, rc1);
    } else { // RF :848-861
        pairOccurrences(rc1, fw2);
        pairOccurrences(rc2, fw1);
    }
    std::vector<std::string> unpaired;
    auto unmappedLine = [&](int r, bool mateMapped, bool mateRev) {
        const char* q = R[r]->qual ? R[r]->qual : "";
        return callText([&](char* o, uint64_t c) { return cmb_sam_unmapped_pe(R[r]->id, R[r]->seq, q, r == 0, mateMapped, mateRev, o, c); });
    };
    auto printSeq = [&](const POcc& p) { return p.o->strand ? R[p.second]->revcomp : R[p.second]->seq; };
    auto printQual = [&](const POcc& p) {
        const cmb_pair_read* rd = R[p.second];
        return p.o->strand ? (rd->revqual ? rd->revqual : "") : (rd->qual ? rd->qual : "");
    };
    // addUnpairedMatches for one read (searchstrategy.cpp:1401-1462; forward occurrences, then reverse-complement ones)
    auto addUnpairedRead = [&](int r, std::vector<POcc>& fw, std::vector<POcc>& rc) {
        std::vector<POcc> temp;
        for (auto* v : {&fw, &rc})
            for (const POcc& p : *v)
                if (p.assigned()) temp.push_back(p);
        fw.clear(), rc.clear();
        if (temp.empty()) {
            if (prm->unmapped_records) unpaired.push_back(unmappedLine(r, false, false));
            return;
        }
        std::stable_sort(temp.begin(), temp.end(), [](const POcc& a, const POcc& b) { return a.o->distance < b.o->distance; });
        const uint32_t best = temp.front().o->distance;
        const uint32_t bestCount = (uint32_t)std::count_if(temp.begin(), temp.end(), [best](const POcc& p) { return p.o->distance == best; });
        bool first = true;
        for (const POcc& p : temp) {
            const cmb_sam_hit h = hitFrom(p, seq_names);
            unpaired.push_back(callText([&](char* o, uint64_t c) {
                return cmb_sam_unpaired(R[r]->id, &h, r == 0, bestCount, best, first, printSeq(p), printQual(p), o, c);
            }));
            first = false;
        }
    };
    auto addUnpairedMatches = [&]() { // searchstrategy.h:1216-1230: the list starts over
        unpaired.clear();
        addUnpairedRead(0, fw1, rc1);
        addUnpairedRead(1, fw2, rc2);
    };
    if (pairs.empty()) { // pairDiscordantly (searchstrategy.cpp:1586-1646)
        const uint64_t m1 = fw1.size() + rc1.size(), m2 = fw2.size() + rc2.size();
        bool done = false;
        if (prm->discordant_allowed && m1 && m2) {
            if (m1 * m2 > 10000) {
                addUnpairedMatches(); // (and once more below, on the emptied lists: what the reference does, :1619-1638)
            } else { // addDiscPairs (:1518-1585)
                auto pairOccs = [&](const POcc& a, const POcc& b) {
                    if (!a.assigned() || !b.assigned()) return;
                    const bool sameRef = a.o->seq_id == b.o->seq_id, aUp = a.o->begin < b.o->begin;
                    const uint32_t frag = sameRef ? (aUp ? b.o->end - a.o->begin : a.o->end - b.o->begin) : 0;
                    pairs.push_back(PPair{aUp ? a : b, aUp ? b : a, true, true, frag, a.o->distance + b.o->distance, true, "", ""});
                };
                for (const POcc& a : fw1) {
                    for (const POcc& b : fw2) pairOccs(a, b);
                    for (const POcc& b : rc2) pairOccs(a, b);
                }
                for (const POcc& a : rc1) {
                    for (const POcc& b : fw2) pairOccs(a, b);
                    for (const POcc& b : rc2) pairOccs(a, b);
                }
                done = !pairs.empty();
            }
        }
        if (!done) {
            if (m1 && m2) addUnpairedMatches();
            else if (!m1 && !m2) { // addBothUnmapped (searchstrategy.h:1236-1247)
                if (prm->unmapped_records)
                    pairs.push_back(PPair{POcc{nullptr, false}, POcc{nullptr, true}, false, false, 0, 0, false, unmappedLine(0, false, false), unmappedLine(1, false, false)});
            } else { // addOneUnmapped (:1463-1517): the mapped read's occurrences, forward ones first, each with the unmapped mate
                const int mr = m1 ? 0 : 1;
                for (int s = 0; s < 2; s++)
                    for (const POcc& p : st[mr][s]) {
                        if (!p.assigned()) continue;
                        pairs.push_back(PPair{p, POcc{nullptr, mr == 0}, true, false, 0, p.o->distance, false, "", unmappedLine(1 - mr, true, p.o->strand != 0)});
                    }
                if (pairs.empty() && prm->unmapped_records)
                    pairs.push_back(PPair{POcc{nullptr, false}, POcc{nullptr, true}, false, false, 0, 0, false, unmappedLine(0, false, false), unmappedLine(1, false, false)});
            }
        }
    }
    // generateSAMPairedEnd (searchstrategy.cpp:1904-1970): the first pair of minimal distance becomes the primary one
    uint32_t nPairs = 0;
    if (!pairs.empty()) {
        size_t mi = 0;
        for (size_t i = 1; i < pairs.size(); i++)
            if (pairs[i].distance < pairs[mi].distance) mi = i;
        const uint32_t bestScore = pairs[mi].distance;
        for (const PPair& p : pairs) nPairs += p.distance == bestScore;
        if (mi != 0) std::swap(pairs[0], pairs[mi]);
        bool primary = true;
        for (PPair& p : pairs) {
            for (int side = 0; side < 2; side++) {
                const POcc& me = side ? p.down : p.up;
                const POcc& mate = side ? p.up : p.down;
                if (!(side ? p.downValid : p.upValid)) continue;
                const cmb_sam_hit h = hitFrom(me, seq_names);
                cmb_sam_hit mh;
                const bool mateValid = side ? p.upValid : p.downValid;
                if (mateValid) mh = hitFrom(mate, seq_names);
                std::string line = callText([&](char* o, uint64_t c) {
                    return cmb_sam_pe(R[me.second]->id, &h, !me.second, mateValid ? &mh : nullptr, nPairs, bestScore, p.fragSize, p.discordant, primary,
                                      printSeq(me), printQual(me), o, c);
                });
                (side ? p.downLine : p.upLine) = line;
            }
            primary = false;
        }
    }
    // OutputWriter::writeChunks (fastq.cpp:662-702)
    std::string text;
    const bool mapped = !pairs.empty() && pairs.front().upValid && pairs.front().downValid;
    const bool mappedHalf = !mapped && !pairs.empty() && (pairs.front().upValid || pairs.front().downValid);
    bool firstWrite = true;
    for (const PPair& p : pairs) {
        text += p.upLine;
        if (!mappedHalf || firstWrite) text += p.downLine;
        firstWrite = false;
    }
    for (const std::string& l : unpaired) text += l;
    if (n_pairs_out) *n_pairs_out = mapped ? (uint32_t)pairs.size() : 0; // TOTAL_UNIQUE_PAIRS
    return putText(text, out, cap);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Inference of the paired-end parameters from pairs whose mates both map unambiguously: addFragmentAndOrientation
// (parallel.cpp:329-360) per pair, then inferPairedEndParameters (:402-466): fragment sizes without the outliers beyond six
// median absolute deviations, mean and standard deviation in single precision as there, max / min insert size = mean -+ 6
// standard deviations, the most frequent orientation.
// ------------------------------------------------------------------------------------------------------------------------------
extern "C" int cmb_pair_infer(const cmb_pair_sample* samples, uint64_t n, cmb_pair_inferred* out) {
    if (!out || (n && !samples)) return failWith(CMB_ERR_INVALID, "bad argument");
    std::memset(out, 0, sizeof(*out));
    out->n_pairs = n;
    if (n == 0) return CMB_OK; // "No pairs mapped unambiguously. Using default values!" (parallel.cpp:408-413): nothing inferred
    std::vector<uint32_t> frag;
    uint64_t cnt[3] = {0, 0, 0};
    for (uint64_t i = 0; i < n; i++) {
        const cmb_pair_sample& s = samples[i];
        const bool firstFirst = s.begin1 < s.begin2;
        frag.push_back(firstFirst ? s.end2 - s.begin1 : s.end1 - s.begin2);
        const bool rc1 = s.strand1 != 0, rc2 = s.strand2 != 0;
        cnt[rc1 == rc2 ? CMB_ORIENTATION_FF : (firstFirst == rc1 ? CMB_ORIENTATION_RF : CMB_ORIENTATION_FR)]++;
    }
    auto median = [](std::vector<uint32_t>& d) -> uint32_t { // calcMedian (:365-372): sorts its argument
        std::sort(d.begin(), d.end());
        return d.size() % 2 == 0 ? (d[d.size() / 2 - 1] + d[d.size() / 2]) / 2 : d[d.size() / 2];
    };
    auto average = [](const std::vector<uint32_t>& v) -> float { // :378-380
        double acc = 0.0;
        for (uint32_t x : v) acc += x;
        return (float)(acc / v.size());
    };
    auto stddev = [](const std::vector<uint32_t>& v, float mean) -> float { // :389-394 (Bessel's correction, float accumulator)
        float accum = 0.0f;
        for (uint32_t x : v) {
            const float d = (float)x;
            accum += (d - mean) * (d - mean);
        }
        return std::sqrt(accum / (v.size() - 1));
    };
    const int32_t med = (int32_t)median(frag);
    std::vector<uint32_t> dev;
    for (uint32_t f : frag) dev.push_back((uint32_t)std::abs((int32_t)f - med));
    const uint32_t mad = median(dev);
    std::vector<uint32_t> kept;
    for (uint32_t f : frag)
        if ((uint32_t)std::abs((int32_t)f - med) < 6u * mad) kept.push_back(f); // PE_STD_DEV_CONSIDERED (definitions.h:54-55)
    float mean = average(kept), sd = stddev(kept, mean);
    if (mean == 0 || sd == 0 || !(mean == mean) || !(sd == sd)) { // (an empty or one-element selection gives NaN there; the reference tests == 0 only)
        mean = average(frag);
        sd = stddev(frag, mean);
    }
    const uint32_t maxDev = (uint32_t)(6 * sd);
    out->mean_insert = mean;
    out->stddev_insert = sd;
    out->max_insert = (uint32_t)(mean + maxDev);
    out->min_insert = mean > maxDev ? (uint32_t)(mean - maxDev) : 0;
    const uint64_t rf = cnt[CMB_ORIENTATION_RF], fr = cnt[CMB_ORIENTATION_FR], ff = cnt[CMB_ORIENTATION_FF];
    out->orientation = (fr >= rf && fr >= ff) ? CMB_ORIENTATION_FR : (rf >= ff) ? CMB_ORIENTATION_RF : CMB_ORIENTATION_FF;
    out->inferred = 1;
    return CMB_OK;
}

// Index construction as product code: FASTA files in, the reference's Vanilla index files out — the
// files `columba` itself and include/columba_amd.hpp (FMIndex) load — and, further down, the run-length compressed flavour's move
// tables, samples, predecessors and PLCP (buildMoveIndex).  Host C++ only; one-off preprocessing, not on the hot path.
//
// Mirrors (reference, src/):
//   preprocessFastaFiles / concatenateAndTransform      buildindex.cpp:150-262, :614-683   (concatenation, upper case, non-ACGT
//                                                       characters replaced with std::minstd_rand(42) — the same engine and
//                                                       distribution, so a libstdc++ build reproduces the reference's text —,
//                                                       or with a seeded pattern: -l)
//   createAndWriteHeaderInfo / writePositionsAndSequenceNames   :341-386             (.headerSN.bin, .pos, .sna, .fsid)
//   writeCharCountsAndCreateAlphabet                    :720-727                          (.cct)
//   createSuffixArray (libsais there)                   :479                              (here: SA-IS, own implementation)
//   generateBWT / createRevBWT / createRevSAWithSanityCheck     :575-585, :706-711, :735-750
//   EncodedText<5>::write                               fmindex/encodedtext.h:93-118, :275 (.bwt: 3 bits per symbol, MSB first)
//   BWTRepresentation<5> / BitvecIntl<4>::index, write  fmindex/bwtrepr.h:56-72, :113; bitvec.h:247-281, :329-349, :378-394 (.brt, .rev.brt)
//   SparseSuffixArray + Bitvec::index, write            fmindex/suffixArray.h:150-164, :229; bitvec.h:134-149, :176-195 (.sa.<s>, .sa.bv.<s>)
//   writeMetaInfo                                       :688-700                          (.meta)
// The suffix array of a text is unique, so the files equal those of the reference's builder byte for byte given the same text
// (tests/test_cpp_builder.py compares them with the files of the harness builder, which tests/test_index_files.py ties to the
// reference's readers).
#pragma once
#include "columba_amd_io.hpp"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

namespace columba_amd {
namespace build {

// ---------------------------------------------------------------- suffix array: SA-IS (Nong, Zhang, Chan 2009)
// s[0..n) over the alphabet [0, K) with s[n - 1] = 0 the unique smallest symbol.  SA receives the suffix array.
namespace sais_detail {
typedef int64_t idx_t;
template <typename C> inline void bucketBounds(const C* s, idx_t n, idx_t K, std::vector<idx_t>& bkt, bool end) {
    std::fill(bkt.begin(), bkt.end(), 0);
    for (idx_t i = 0; i < n; i++) bkt[(size_t)s[i]]++;
    idx_t sum = 0;
    for (idx_t c = 0; c < K; c++) {
        sum += bkt[(size_t)c];
        bkt[(size_t)c] = end ? sum : sum - bkt[(size_t)c];
    }
}
template <typename C> void induce(const C* s, idx_t* SA, idx_t n, idx_t K, const std::vector<uint8_t>& isS, std::vector<idx_t>& bkt) {
    bucketBounds(s, n, K, bkt, false); // L-type suffixes: left to right, into bucket starts
    for (idx_t i = 0; i < n; i++) {
        const idx_t j = SA[i] - 1;
        if (SA[i] > 0 && !isS[(size_t)j]) SA[bkt[(size_t)s[j]]++] = j;
    }
    bucketBounds(s, n, K, bkt, true); // S-type suffixes: right to left, into bucket ends
    for (idx_t i = n - 1; i >= 0; i--) {
        const idx_t j = SA[i] - 1;
        if (SA[i] > 0 && isS[(size_t)j]) SA[--bkt[(size_t)s[j]]] = j;
    }
}
template <typename C> void sais(const C* s, idx_t* SA, idx_t n, idx_t K) {
    if (n == 1) {
        SA[0] = 0;
        return;
    }
    std::vector<uint8_t> isS((size_t)n, 0);
    isS[(size_t)n - 1] = 1;
    for (idx_t i = n - 2; i >= 0; i--) isS[(size_t)i] = s[i] < s[i + 1] || (s[i] == s[i + 1] && isS[(size_t)i + 1]);
    auto isLMS = [&](idx_t i) { return i > 0 && isS[(size_t)i] && !isS[(size_t)i - 1]; };
    std::vector<idx_t> bkt((size_t)K);
    // stage 1: sort the LMS substrings
    bucketBounds(s, n, K, bkt, true);
    std::fill(SA, SA + n, (idx_t)-1);
    for (idx_t i = 1; i < n; i++)
        if (isLMS(i)) SA[--bkt[(size_t)s[i]]] = i;
    induce(s, SA, n, K, isS, bkt);
    // compact the sorted LMS substrings into the first n1 entries
    idx_t n1 = 0;
    for (idx_t i = 0; i < n; i++)
        if (isLMS(SA[i])) SA[n1++] = SA[i];
    std::fill(SA + n1, SA + n, (idx_t)-1);
    // name them
    idx_t name = 0, prev = -1;
    for (idx_t i = 0; i < n1; i++) {
        const idx_t pos = SA[i];
        bool diff = prev < 0;
        if (!diff)
            for (idx_t d = 0;; d++) {
                if (s[pos + d] != s[prev + d] || isS[(size_t)(pos + d)] != isS[(size_t)(prev + d)]) {
                    diff = true;
                    break;
                }
                if (d > 0 && (isLMS(pos + d) || isLMS(prev + d))) break; // both substrings ended together
            }
        if (diff) {
            name++;
            prev = pos;
        }
        SA[n1 + pos / 2] = name - 1;
    }
    for (idx_t i = n - 1, j = n - 1; i >= n1; i--)
        if (SA[i] >= 0) SA[j--] = SA[i];
    // stage 2: the reduced problem
    idx_t* SA1 = SA;
    idx_t* s1 = SA + n - n1;
    if (name < n1) {
        std::vector<idx_t> s1copy(s1, s1 + n1); // (the recursion writes SA1 = SA[0, n1), which does not overlap s1, but keep it simple)
        sais<idx_t>(s1copy.data(), SA1, n1, name);
    } else {
        for (idx_t i = 0; i < n1; i++) SA1[s1[i]] = i;
    }
    // stage 3: induce the suffix array from the sorted LMS suffixes
    bucketBounds(s, n, K, bkt, true);
    for (idx_t i = 1, j = 0; i < n; i++)
        if (isLMS(i)) s1[j++] = i; // positions of the LMS suffixes in text order
    for (idx_t i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
    std::fill(SA + n1, SA + n, (idx_t)-1);
    for (idx_t i = n1 - 1; i >= 0; i--) {
        const idx_t j = SA[i];
        SA[i] = -1;
        SA[--bkt[(size_t)s[j]]] = j;
    }
    induce(s, SA, n, K, isS, bkt);
}
} // namespace sais_detail

// the suffix array of an arbitrary byte string (a shorter suffix that is a prefix of a longer one sorts first, as libsais does)
inline std::vector<uint32_t> suffixArray(const std::string& T) {
    const int64_t n = (int64_t)T.size();
    std::vector<uint16_t> s((size_t)n + 1);
    for (int64_t i = 0; i < n; i++) s[(size_t)i] = (uint16_t)((unsigned char)T[(size_t)i] + 1);
    s[(size_t)n] = 0; // virtual sentinel
    std::vector<int64_t> SA((size_t)n + 1);
    sais_detail::sais<uint16_t>(s.data(), SA.data(), n + 1, 257);
    std::vector<uint32_t> out((size_t)n);
    for (int64_t i = 0; i < n; i++) out[(size_t)i] = (uint32_t)SA[(size_t)i + 1]; // (SA[0] is the sentinel)
    return out;
}

// ---------------------------------------------------------------- FASTA -> text
struct Text {
    std::string T;                       // upper case ACGT, '$' at the end
    std::vector<uint32_t> positions;     // start of every sequence + the end of the last one
    std::vector<std::string> seqNames;
    std::vector<uint32_t> firstSeqIDPerFile;
};

inline Text preprocessFastaFiles(const std::vector<std::string>& fastaFiles, uint32_t seedLength) {
    std::minstd_rand gen(42);
    auto randomACGT = [&gen](char c) -> char {
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') {
            std::uniform_int_distribution<size_t> distribution(0, 3);
            return "ACGT"[distribution(gen)];
        }
        return c;
    };
    std::string seed;
    for (uint32_t i = 0; i < seedLength; i++) seed += randomACGT('N');
    gen.seed(42);
    std::function<char(char, size_t&)> replace;
    if (seedLength == 0)
        replace = [&](char c, size_t&) { return randomACGT(c); };
    else
        replace = [&](char c, size_t& seedIndex) -> char {
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T') {
                const char r = seed[seedIndex];
                seedIndex = (seedIndex + 1) % seed.length();
                return r;
            }
            seedIndex = 0;
            return c;
        };
    Text out;
    for (const std::string& file : fastaFiles) {
        LineSource in(file);
        out.firstSeqIDPerFile.push_back((uint32_t)out.seqNames.size());
        std::string sequence, line;
        bool sequenceName = false;
        size_t startPosition = out.T.size(), seedIndex = 0;
        auto flush = [&]() {
            for (char c : sequence) out.T += replace((char)std::toupper((unsigned char)c), seedIndex);
            out.positions.push_back((uint32_t)startPosition);
            startPosition += sequence.size();
            sequence.clear();
        };
        while (in.getline(line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            if (line[0] == '>') {
                if (!sequence.empty()) flush();
                const std::string description = line.substr(1);
                const std::string name = description.substr(0, description.find(' '));
                if (std::find(out.seqNames.begin(), out.seqNames.end(), name) != out.seqNames.end())
                    throw std::runtime_error("Error: Sequence name " + name + " in file " + file + " is not unique!");
                out.seqNames.push_back(name);
                sequenceName = true;
            } else {
                sequence += line;
            }
        }
        flush(); // the last sequence of the file
        if (!sequenceName) out.seqNames.push_back(file);
        if (out.T.size() >= 0xFFFFFFFEull) throw std::runtime_error("the text does not fit a 32-bit length_t");
    }
    out.positions.push_back((uint32_t)out.T.size());
    if (out.T.empty() || out.T.back() != '$') out.T += '$';
    return out;
}

// ---------------------------------------------------------------- arrays in the reference's layouts
inline uint32_t codeOf(char c) { return c == '$' ? 0u : c == 'A' ? 1u : c == 'C' ? 2u : c == 'G' ? 3u : 4u; }

// rank9-style counts over 64-bit words: per block of 8 words {absolute count before the block, seven 9-bit partial sums}
inline void rankCounts(const std::vector<uint64_t>& words, size_t stride, size_t offset, size_t nWords, size_t nBlocks, std::vector<uint64_t>& l1,
                       std::vector<uint64_t>& l2) {
    l1.assign(nBlocks, 0);
    l2.assign(nBlocks, 0);
    uint64_t total = 0;
    for (size_t b = 0; b < nBlocks; b++) {
        l1[b] = total;
        uint64_t within = 0;
        for (size_t j = 0; j < 8; j++) {
            const size_t w = b * 8 + j;
            if (j > 0 && w < nWords) l2[b] |= within << (9 * (j - 1)); // (words past the end are never visited: their partial sums stay 0)
            if (w < nWords) within += (uint64_t)__builtin_popcountll(words[w * stride + offset]);
        }
        total += within;
    }
}

struct BwtBitvectors { // BWTRepresentation<5>: four cumulative bitvectors of n + 1 bits, interleaved, with their counts
    uint64_t dollarPos = 0, N = 0;
    std::vector<uint64_t> bv, counts;
};
inline BwtBitvectors bwtBitvectors(const std::string& bwt) {
    BwtBitvectors r;
    const size_t n = bwt.size();
    r.N = n + 1;
    const size_t nw = (r.N + 63) / 64, nblk = (r.N + 511) / 512;
    r.dollarPos = n;
    for (size_t i = 0; i < n; i++)
        if (bwt[i] == '$') {
            r.dollarPos = i;
            break;
        }
    r.bv.assign(nw * 4, 0);
    for (size_t i = 0; i < n; i++) {
        const uint32_t c = codeOf(bwt[i]);
        if (c == 0) continue;
        for (uint32_t k = c; k <= 4; k++) r.bv[(i / 64) * 4 + (k - 1)] |= 1ull << (i % 64); // bitvector k: symbols 1 .. k
    }
    r.counts.assign(nblk * 8, 0);
    std::vector<uint64_t> l1, l2;
    for (uint32_t k = 0; k < 4; k++) {
        rankCounts(r.bv, 4, k, nw, nblk, l1, l2);
        for (size_t b = 0; b < nblk; b++) {
            r.counts[(b * 4 + k) * 2] = l1[b];
            r.counts[(b * 4 + k) * 2 + 1] = l2[b];
        }
    }
    return r;
}

struct SparseSA {
    std::vector<uint64_t> bv, counts; // Bitvec over the rows (sampled or not) with its rank9 counts
    std::vector<uint32_t> samples;    // the sampled values in row order
};
inline SparseSA sparseSuffixArray(const std::vector<uint32_t>& SA, uint32_t sparseness) {
    SparseSA r;
    const size_t n = SA.size(), nw = (n + 63) / 64, nblk = (nw + 7) / 8;
    r.bv.assign(nw, 0);
    for (size_t i = 0; i < n; i++)
        if (SA[i] % sparseness == 0) {
            r.bv[i / 64] |= 1ull << (i % 64);
            r.samples.push_back(SA[i]);
        }
    std::vector<uint64_t> l1, l2;
    rankCounts(r.bv, 1, 0, nw, nblk, l1, l2);
    const size_t cw = (nw + 7) / 4;
    r.counts.assign(cw, 0);
    for (size_t b = 0; b < nblk; b++) {
        if (2 * b < cw) r.counts[2 * b] = l1[b];
        if (2 * b + 1 < cw) r.counts[2 * b + 1] = l2[b];
    }
    return r;
}

inline std::vector<uint64_t> encodeBwt(const std::string& bwt) { // 3 bits per symbol, most significant bit first, a continuous stream
    const size_t n = bwt.size(), nw = (n * 3) / 64 + 1;
    std::vector<uint64_t> w(nw, 0);
    for (size_t i = 0; i < n; i++) {
        const uint32_t c = codeOf(bwt[i]);
        for (uint32_t b = 0; b < 3; b++) {
            const size_t bit = 3 * i + b;
            if ((c >> (2 - b)) & 1u) w[bit / 64] |= 1ull << (63 - bit % 64);
        }
    }
    return w;
}

// ---------------------------------------------------------------- files
template <typename T> inline void put(std::ofstream& f, const T& v) { f.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <typename T> inline void putAll(std::ofstream& f, const std::vector<T>& v) {
    if (!v.empty()) f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}
inline std::ofstream openOut(const std::string& name) {
    std::ofstream f(name, std::ios::binary);
    if (!f) throw std::runtime_error("Cannot open file for writing: " + name);
    return f;
}

// FASTA files -> <base>.{meta, cct, txt.bin, bwt, brt, rev.brt, sa.<s>, sa.bv.<s>, pos, sna, fsid, headerSN.bin}
// allSparsenessFactors: the sparse suffix array for every factor 1, 2, 4 ... 128 (`-a`, buildindex.cpp:1914-1918) instead of the one given
inline void buildIndex(const std::vector<std::string>& fastaFiles, const std::string& base, uint32_t sparseness = 4, uint32_t seedLength = 0,
                       bool allSparsenessFactors = false) {
    if (sparseness == 0 || (sparseness & (sparseness - 1)) != 0) throw std::runtime_error("the sparseness factor must be a power of two");
    const Text tx = preprocessFastaFiles(fastaFiles, seedLength);
    const std::string& T = tx.T;
    const size_t n = T.size();
    { // text, sequences, header lines
        std::ofstream f = openOut(base + ".txt.bin");
        put<uint32_t>(f, (uint32_t)n);
        f.write(T.data(), (std::streamsize)n);
        std::ofstream h = openOut(base + ".headerSN.bin");
        for (size_t i = 0; i < tx.seqNames.size(); i++)
            h << "@SQ\tSN:" << tx.seqNames[i] << "\tLN:" << tx.positions[i + 1] - tx.positions[i] << "\n";
        std::ofstream p = openOut(base + ".pos");
        putAll(p, tx.positions);
        std::ofstream s = openOut(base + ".sna");
        for (const std::string& name : tx.seqNames) {
            put<uint64_t>(s, (uint64_t)name.size());
            s.write(name.data(), (std::streamsize)name.size());
        }
        std::ofstream fs = openOut(base + ".fsid");
        putAll(fs, tx.firstSeqIDPerFile);
    }
    { // character counts
        std::vector<uint32_t> cct(256, 0);
        for (char c : T) cct[(unsigned char)c]++;
        std::ofstream f = openOut(base + ".cct");
        putAll(f, cct);
    }
    { // the text's own BWT: 3-bit BWT, bitvectors, sparse suffix array
        const std::vector<uint32_t> SA = suffixArray(T);
        std::string bwt(n, '$');
        for (size_t i = 0; i < n; i++) bwt[i] = SA[i] > 0 ? T[SA[i] - 1] : T.back();
        for (uint32_t sf = allSparsenessFactors ? 1u : sparseness; sf <= (allSparsenessFactors ? 128u : sparseness); sf *= 2) {
            const SparseSA ssa = sparseSuffixArray(SA, sf);
            std::ofstream f = openOut(base + ".sa.bv." + std::to_string(sf));
            put<uint64_t>(f, (uint64_t)n);
            putAll(f, ssa.bv);
            putAll(f, ssa.counts);
            std::ofstream g = openOut(base + ".sa." + std::to_string(sf));
            putAll(g, ssa.samples);
        }
        const std::vector<uint64_t> words = encodeBwt(bwt);
        std::ofstream f = openOut(base + ".bwt");
        put<uint64_t>(f, (uint64_t)n);
        put<uint64_t>(f, (uint64_t)words.size());
        putAll(f, words);
        const BwtBitvectors b = bwtBitvectors(bwt);
        std::ofstream g = openOut(base + ".brt");
        put<uint64_t>(g, b.dollarPos);
        put<uint64_t>(g, b.N);
        putAll(g, b.bv);
        putAll(g, b.counts);
    }
    { // the reversed text (the '$' in front: createRevSAWithSanityCheck reverses the whole string)
        std::string revT(T.rbegin(), T.rend());
        const std::vector<uint32_t> revSA = suffixArray(revT);
        revT.clear();
        std::string rbwt(n, '$');
        for (size_t i = 0; i < n; i++) rbwt[i] = revSA[i] > 0 ? T[n - revSA[i]] : T.front();
        const BwtBitvectors b = bwtBitvectors(rbwt);
        std::ofstream g = openOut(base + ".rev.brt");
        put<uint64_t>(g, b.dollarPos);
        put<uint64_t>(g, b.N);
        putAll(g, b.bv);
        putAll(g, b.counts);
    }
    std::ofstream m = openOut(base + ".meta");
    m << 21 << "\n" << 4 << "\n" << "VANILLA" << "\n";
}

// ---------------------------------------------------------------- the run-length compressed flavour (b-move)
// Mirrors (reference, src/, RUN_LENGTH_COMPRESSION build with its 64-bit length_t):
//   processFastaFiles                                   buildindex.cpp:2008-2029   (seed length 100 by default, definitions.h:41; the text is
//                                                       not written: noWriting)
//   createIndex (RLC)                                   :1606-1687                  (PLCP, samples at run boundaries, predecessors, move tables
//                                                       of the text and of the reversed text)
//   buildSamples / processSamplesAndPreds               :950-1013, :1044-1066, :1540-1603
//   createAndWriteMove + MoveLFReprBP::write            :826-915, bmove/moverepr.cpp:145-181, :75-77 (.LFBP: n, r, position of '$', then r + 1 rows of
//                                                       ceil((3 + 2 ceil(log2 n) + ceil(log2 r)) / 8) bytes: character, run start, LF of the run
//                                                       start and the run that holds it, every value cut to its field)
//   PLCP (Kasai et al. through phi)                     bmove/plcp.h:56-80
// The reference keeps samples, predecessors and PLCP in sdsl containers whose serialisation belongs to sdsl (absent here): their CONTENTS are
// written as plain little-endian 64-bit arrays, the files include/columba_amd_bmove.hpp (BMove) reads — <base>.smpf.u64 .smpl.u64
// .rev.smpf.u64 .rev.smpl.u64 .prdf.u64 .ftr.u64 .prdl.u64 .ltr.u64 .plcp.pos.u64 .plcp.sum.u64 — beside the two .LFBP files, which are
// the reference's own format.  tests/test_cpp_builder.py compares every file with the harness builder's (columba_amd/movebuild.py).
inline unsigned bitsFor(uint64_t v) { // ceil(log2(v)), moverepr.h:44-46
    unsigned b = 0;
    while (b < 64 && (1ull << b) < v) b++;
    return b;
}

inline std::vector<uint8_t> packLFBP(const std::string& bwt) {
    const uint64_t n = bwt.size();
    if ((n & (n - 1)) == 0) // moverepr.cpp:75-77: the terminating row's start position n does not fit ceil(log2 n) bits
        throw std::runtime_error("a text size that is a power of two cannot be packed into a move table");
    uint64_t cnt[5] = {0, 0, 0, 0, 0}, cum[5], zeroPos = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t c = codeOf(bwt[i]);
        if (c == 0) zeroPos = i;
        cnt[c]++;
    }
    cum[0] = 0;
    for (int c = 1; c < 5; c++) cum[c] = cum[c - 1] + cnt[c - 1];
    std::vector<uint64_t> starts, lf;
    std::vector<uint8_t> head;
    uint64_t seen[5] = {0, 0, 0, 0, 0};
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t c = codeOf(bwt[i]);
        if (i == 0 || bwt[i] != bwt[i - 1]) {
            starts.push_back(i);
            head.push_back((uint8_t)c);
            lf.push_back(cum[c] + seen[c]); // LF of the run's first position
        }
        seen[c]++;
    }
    const uint64_t r = starts.size();
    const unsigned bitsN = bitsFor(n), bitsR = bitsFor(r), bitsC = 3;
    const unsigned totalBytes = (bitsC + 2 * bitsN + bitsR + 7) / 8;
    const uint64_t maskN = bitsN >= 64 ? ~0ull : (1ull << bitsN) - 1, maskR = bitsR >= 64 ? ~0ull : (1ull << bitsR) - 1;
    std::vector<uint8_t> out(24 + (size_t)(r + 1) * totalBytes, 0);
    const uint64_t hdr[3] = {n, r, zeroPos};
    std::memcpy(out.data(), hdr, 24);
    auto place = [](unsigned __int128& row, uint64_t v, unsigned off) { row |= (unsigned __int128)v << off; };
    for (uint64_t i = 0; i <= r; i++) {
        uint64_t c, a, b, d;
        if (i < r) {
            c = head[i], a = starts[i], b = lf[i];
            d = (uint64_t)(std::upper_bound(starts.begin(), starts.end(), b) - starts.begin()) - 1; // the run that holds the LF position
        } else
            c = 0, a = n, b = n, d = r; // the terminating row
        unsigned __int128 row = 0;
        place(row, c, 0);
        place(row, a & maskN, bitsC);
        place(row, b & maskN, bitsC + bitsN);
        place(row, d & maskR, bitsC + 2 * bitsN);
        std::memcpy(out.data() + 24 + (size_t)i * totalBytes, &row, totalBytes);
    }
    return out;
}

// suffix-array values at the first and the last position of every BWT run (buildSamples)
inline void runSamples(const std::vector<uint32_t>& SA, const std::string& bwt, std::vector<uint64_t>& first, std::vector<uint64_t>& last) {
    first.clear(), last.clear();
    for (size_t i = 0; i < bwt.size(); i++) {
        if (i == 0 || bwt[i] != bwt[i - 1]) first.push_back(SA[i]);
        if (i + 1 == bwt.size() || bwt[i] != bwt[i + 1]) last.push_back(SA[i]);
    }
}

// predecessor structure of phi / phi^-1: the text positions (sample - 1, cyclically) in increasing order and the run each belongs to
inline void predecessors(const std::vector<uint64_t>& samples, uint64_t n, std::vector<uint64_t>& marked, std::vector<uint64_t>& toRun) {
    std::vector<std::pair<uint64_t, uint64_t>> kv(samples.size());
    for (size_t i = 0; i < samples.size(); i++) kv[i] = {samples[i] > 0 ? samples[i] - 1 : n - 1, (uint64_t)i};
    std::sort(kv.begin(), kv.end());
    marked.resize(kv.size()), toRun.resize(kv.size());
    for (size_t i = 0; i < kv.size(); i++) marked[i] = kv[i].first, toRun[i] = kv[i].second;
}

// PLCP[p] = longest common prefix of the suffix at p and its predecessor in suffix-array order, in the run-length form cmb_move_desc takes:
// the positions q with PLCP[q] != PLCP[q - 1] - 1 (0 among them) and PLCP[q] + q there
inline void plcpRuns(const std::string& T, const std::vector<uint32_t>& SA, std::vector<uint64_t>& pos, std::vector<uint64_t>& sum) {
    const size_t n = T.size();
    std::vector<uint32_t> phi(n);
    const uint32_t none = 0xFFFFFFFFu;
    phi[SA[0]] = none;
    for (size_t i = 1; i < n; i++) phi[SA[i]] = SA[i - 1];
    pos.clear(), sum.clear();
    size_t l = 0;
    uint64_t prev = 0;
    for (size_t p = 0; p < n; p++) {
        if (phi[p] == none)
            l = 0;
        else {
            const size_t q = phi[p];
            while (p + l < n && q + l < n && T[p + l] == T[q + l]) l++;
        }
        if (p == 0 || (uint64_t)l + 1 != prev) pos.push_back(p), sum.push_back((uint64_t)l + p);
        prev = l;
        if (l > 0) l--;
    }
}

inline void put64(const std::string& name, const std::vector<uint64_t>& v) {
    std::ofstream f = openOut(name);
    putAll(f, v);
}

// FASTA files -> <base>.{meta, cct, pos, sna, fsid, headerSN.bin, LFBP, rev.LFBP, *.u64} (64-bit length_t as the reference's RLC build);
// keepText: also <base>.txt.bin — the reference's RLC flavour does not keep the text, this framework computes alignments of b-move
// occurrences on it (BMove::attachText)
inline void buildMoveIndex(const std::vector<std::string>& fastaFiles, const std::string& base, uint32_t seedLength = 100, bool keepText = false) {
    const Text tx = preprocessFastaFiles(fastaFiles, seedLength);
    const std::string& T = tx.T;
    const size_t n = T.size();
    {
        if (keepText) {
            std::ofstream f = openOut(base + ".txt.bin");
            put<uint64_t>(f, (uint64_t)n);
            f.write(T.data(), (std::streamsize)n);
        }
        std::ofstream h = openOut(base + ".headerSN.bin");
        for (size_t i = 0; i < tx.seqNames.size(); i++)
            h << "@SQ\tSN:" << tx.seqNames[i] << "\tLN:" << tx.positions[i + 1] - tx.positions[i] << "\n";
        put64(base + ".pos", std::vector<uint64_t>(tx.positions.begin(), tx.positions.end()));
        std::ofstream s = openOut(base + ".sna");
        for (const std::string& name : tx.seqNames) {
            put<uint64_t>(s, (uint64_t)name.size());
            s.write(name.data(), (std::streamsize)name.size());
        }
        put64(base + ".fsid", std::vector<uint64_t>(tx.firstSeqIDPerFile.begin(), tx.firstSeqIDPerFile.end()));
        std::vector<uint64_t> cct(256, 0);
        for (char c : T) cct[(unsigned char)c]++;
        put64(base + ".cct", cct);
    }
    auto writeBytes = [](const std::string& name, const std::vector<uint8_t>& b) {
        std::ofstream f = openOut(name);
        putAll(f, b);
    };
    {
        const std::vector<uint32_t> SA = suffixArray(T);
        std::string bwt(n, '$');
        for (size_t i = 0; i < n; i++) bwt[i] = SA[i] > 0 ? T[SA[i] - 1] : T.back();
        std::vector<uint64_t> a, b, c, d;
        plcpRuns(T, SA, a, b);
        put64(base + ".plcp.pos.u64", a), put64(base + ".plcp.sum.u64", b);
        runSamples(SA, bwt, a, b);
        put64(base + ".smpf.u64", a), put64(base + ".smpl.u64", b);
        predecessors(a, n, c, d);
        put64(base + ".prdf.u64", c), put64(base + ".ftr.u64", d);
        predecessors(b, n, c, d);
        put64(base + ".prdl.u64", c), put64(base + ".ltr.u64", d);
        writeBytes(base + ".LFBP", packLFBP(bwt));
    }
    {
        std::string revT(T.rbegin(), T.rend());
        const std::vector<uint32_t> revSA = suffixArray(revT);
        revT.clear();
        std::string rbwt(n, '$');
        for (size_t i = 0; i < n; i++) rbwt[i] = revSA[i] > 0 ? T[n - revSA[i]] : T.front();
        std::vector<uint64_t> a, b;
        runSamples(revSA, rbwt, a, b);
        put64(base + ".rev.smpf.u64", a), put64(base + ".rev.smpl.u64", b);
        writeBytes(base + ".rev.LFBP", packLFBP(rbwt));
    }
    std::ofstream m = openOut(base + ".meta");
    m << 21 << "\n" << 8 << "\n" << "RLC" << "\n";
}

} // namespace build
} // namespace columba_amd

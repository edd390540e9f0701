// Device-side search: shared definitions — strategy tables, work items, queues.
//
// The reference walks one read x strand through
//   SearchStrategy::matchWithSearches   reference src/searchstrategy.cpp:425-493
// i.e. partitioning (:141-419), part-level in-text pre-verification (:464-476), dynamic scheme
// selection (src/searchstrategy.h:2505-2537), doRecSearch (:1181-1254) and the recursive searches
//   IndexInterface::recApproxMatchEdit / branchAndBound / goDeeper   src/indexinterface.cpp:377-669
//   IndexInterface::recApproxMatchHamming                           src/indexinterface.cpp:1211-1304
// on one thread.  Here every stage is its own kernel (dev_partition.hpp, dev_bfs_edit.hpp,
// dev_bfs_hamming.hpp, kernels.hpp) and the stages talk through global queues of small records.
//
// What the reference does inline — locate (findSA), in-text verification, FM-occurrence
// conversion — is NOT done by the search kernels: they only emit compact work items, and dedicated
// regular kernels (kernels.hpp) consume them.  The search never depends on the result of a
// verification, so the occurrence SET is unchanged.
#pragma once
#include "dev_index.hpp"
#include "dev_matrix.hpp"

namespace cmb {

constexpr int MAXP = 8;      // max parts of the COMMON tables (k <= 7: k + 1 parts; k + 2 for kuch2 / 01*0 up to k = 4 ...)
constexpr int MAXP_WIDE = 16; // ... and of the wide ones (the greedy schemes for 8 ... 13 errors have k + 1 parts: searchstrategy.h:3396-3658)
constexpr int MAXS = 16;     // max searches per scheme
constexpr int MAXSCH = 4;    // max alternative schemes per k (dynamic selection)
constexpr int MAX_READ = 480; // (9-bit part bounds and matrix dimensions in the records: < 512 with the band; contexts of batches with reads beyond
                              // 320 characters cache the match words of 512 rows instead of 352, dev_bfs_edit.hpp: ctxGeometry)
constexpr int DESC_MAX = 56; // descendants handed to the next phase

// ---- strategy tables (built on the host by host/schemes.cpp) -------------------------------
// Templates over the number of parts the tables hold: the kernels of the headline path are instances for MAXP (their registers,
// LDS and record sizes are those of that size), batches whose scheme has more parts run instances for MAXP_WIDE.
template <int MP> struct DevSearchT { // Search, src/search.h:55-101
    uint8_t n;
    uint8_t order[MP], L[MP], U[MP], dir[MP], dsw[MP];
    uint8_t low[MP], high[MP]; // lowestAndHighestPartsProcessedBefore[i]
    uint8_t uniAll, uniIdx;    // isUnidirectionalBackwards(i) = uniAll || i >= uniIdx (:479)
};
template <int MP> struct DevSchemeT {
    uint8_t nSearches, critical; // SearchScheme::criticalPartIndex (search.h:525)
    DevSearchT<MP> s[MAXS];
};
template <int MP> struct DevStrategyKT { // everything matchWithSearches needs for one distance k
    uint8_t metric, partition, numParts, nSchemes;
    uint32_t kmerCutOff;
    double seeding[MP];   // getSeedingPositions (searchstrategy.h:1825)
    uint64_t weights[MP]; // getWeights (:283)
    double begins[MP];    // getBegins (:245)
    DevSchemeT<MP> sch[MAXSCH];
};
typedef DevSearchT<MAXP> DevSearch;
typedef DevSchemeT<MAXP> DevScheme;
typedef DevStrategyKT<MAXP> DevStrategyK;

// ---- work items -----------------------------------------------------------------------------
enum { ITEM_EDIT = 0, ITEM_HAMMING = 1, ITEM_EXACT = 2 };
// item = {rsId, saRow, startDiff | lengthBefore | remaining, meta}
// meta: shift[0:12) maxED[12:16) minED[16:20) fixed[20] kind[21:23)
__device__ __forceinline__ uint32_t packMeta(uint32_t shift, uint32_t maxED, uint32_t minED, uint32_t fixed,
                                             uint32_t kind) {
    return (shift & 0xFFFu) | (maxED << 12) | (minED << 16) | (fixed << 20) | (kind << 21);
}
struct FMOccRec { // in-index occurrence (FMOcc, src/indexhelpers.h:1353)
    uint32_t rsId, b, e, depth, dist, shift;
};
struct TextOccRec { // in-text occurrence before filtering
    uint32_t rsId, begin, end, dist;
};

// every bit of the batch's flag word (Queues::cnt[3]), one enum so that no two meanings share a bit
enum : uint32_t { FLAG_ITEM_OVERFLOW = 1, FLAG_FMOCC_OVERFLOW = 2, FLAG_TEXT_OVERFLOW = 4, FLAG_CAPACITY = 8,
                  FLAG_UNSUPPORTED_READ = 16, FLAG_DFS_OVERFLOW = 32, FLAG_TRACE_RULE = 64,
                  // frontier pools (dev_bfs_edit.hpp / dev_bfs_hamming.hpp)
                  FLAG_BFS_Q = 128, FLAG_BFS_EV = 256, FLAG_BFS_F = 512, FLAG_BFS_CTX = 1024, FLAG_BFS_ARENA = 2048,
                  // dynamic partitioning: the initial seeds of a read overlap (the reference asserts they do not,
                  // searchstrategy.cpp:404-407, and its CLI caps the k-mer size accordingly, alignparameters.cpp:1070-1114)
                  FLAG_SEED_OVERLAP = 4096,
                  // naive backtracking (dev_bfs_naive.hpp): its node queue
                  FLAG_NAIVE_Q = 8192,
                  // a phase whose first column does not fit the 32-bit in-index matrix (dev_bfs_edit.hpp: GeoN32): the batch re-runs on GeoN
                  FLAG_NARROW_MATRIX = 16384 };
constexpr uint32_t FLAG_BITS[] = {FLAG_ITEM_OVERFLOW, FLAG_FMOCC_OVERFLOW, FLAG_TEXT_OVERFLOW, FLAG_CAPACITY,
                                  FLAG_UNSUPPORTED_READ, FLAG_DFS_OVERFLOW, FLAG_TRACE_RULE, FLAG_BFS_Q, FLAG_BFS_EV,
                                  FLAG_BFS_F, FLAG_BFS_CTX, FLAG_BFS_ARENA, FLAG_SEED_OVERLAP, FLAG_NAIVE_Q, FLAG_NARROW_MATRIX};
constexpr bool flagBitsDisjoint() {
    uint32_t seen = 0;
    for (uint32_t b : FLAG_BITS) {
        if (b == 0 || (b & (b - 1)) != 0 || (seen & b) != 0) return false;
        seen |= b;
    }
    return true;
}
static_assert(flagBitsDisjoint(), "every flag is one bit of its own");

// The layouts of the low bits of a filter key (`layout`): 0 edit distance up to 7 (begin << 8 | distance[7:5] | width - (len - k) [4:1] |
// strand), 1 Hamming distance (distance[7:4], the width is the read's), 2 edit distance 8 ... 13 (begin << 10 | distance[9:6] |
// width - (len - k) [5:1] | strand; 22 bits are left for the group: sub-batches of at most 2^21 reads).
struct KeyBits {
    uint32_t group, begin, dist, distMask, wMask;
};
__host__ __device__ __forceinline__ KeyBits keyBits(uint32_t layout) {
    return layout == 2u ? KeyBits{42u, 10u, 6u, 15u, 31u} : layout == 1u ? KeyBits{40u, 8u, 4u, 15u, 0u} : KeyBits{40u, 8u, 5u, 7u, 15u};
}

struct Queues {
    uint4* items;
    uint32_t itemCap;
    FMOccRec* fm;
    uint32_t fmCap;
    TextOccRec* text;
    uint32_t textCap;
    uint32_t* cnt; // [0] items, [1] fm, [2] text, [3] flags, [5] search tasks, [7] traceback tasks
    unsigned long long* counters; // CMB_CNT_MAX
};

} // namespace cmb

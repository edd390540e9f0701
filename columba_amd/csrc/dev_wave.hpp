// Wavefront-level helpers shared by the kernels of both index backends (64-lane wavefronts, gfx950): counter flushes,
// prefix sums, queue appends with one atomic per wavefront, per-wavefront queue chunks.
#pragma once
#include "dev_search.hpp"

namespace cmb {

// Per-lane counters are summed over the wavefront first (all 64 lanes call this, at the end of a kernel):
// one atomic per wavefront and counter instead of one per lane.
__device__ __forceinline__ void flushCounters(const Queues& q, const uint32_t* local, const int* which, int n) {
    for (int i = 0; i < n; i++) {
        unsigned long long v = local[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&q.counters[which[i]], v);
    }
}

// wave-wide exclusive prefix sum (all 64 lanes must call)
__device__ __forceinline__ uint32_t waveExclusiveScan(uint32_t v, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if ((int)lane >= d) x += y;
    }
    total = __shfl(x, 63);
    return x - v;
}
// the same on the cross-lane data path (DPP row shifts and row broadcasts instead of six LDS permutes): INCLUSIVE prefix sum over
// the 64 lanes; all lanes must be active
__device__ __forceinline__ uint32_t waveInclusiveScanDpp(uint32_t v) {
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); // row_shr:1 (rows of 16 lanes; nothing shifted in: 0)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2 and 3
    return x;
}
__device__ __forceinline__ uint32_t waveLastLane(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, 63); }

// one atomic per wavefront: returns this lane's first slot for its `n` records in a queue
__device__ __forceinline__ uint32_t waveAppend(uint32_t* counter, uint32_t n, uint32_t& total) {
    const uint32_t off = waveExclusiveScan(n, total);
    uint32_t base = 0;
    if (total) {
        if ((threadIdx.x & 63u) == 0) base = atomicAdd(counter, total);
        base = __shfl(base, 0);
    }
    return base + off;
}

// Queue space in per-wavefront CHUNKS.  Atomics on one address are served at ~90 per microsecond by the
// L2 atomic unit, whichever wavefront issues them: a kernel whose wavefronts append a few records every
// loop iteration is bound by that (2 M appends = 22 ms) and does not get faster with more wavefronts.  So a
// wavefront reserves `chunk` slots with ONE atomic and hands them out locally (prefix sum); what is left of
// a chunk when it is retired, or when the kernel ends, is filled with HOLES (records whose first word is
// 0xFFFFFFFF), which the consumers skip.  All members are wave-uniform.
struct WaveChunk {
    uint32_t base = 0, used = 0, size = 0;
    // slots [returned, returned + n) for this lane's n records; 0xFFFFFFFF if the queue overflowed.
    // All 64 lanes must call; `hole(i)` writes a hole at slot i.
    template <class Hole>
    __device__ __forceinline__ uint32_t alloc(uint32_t* counter, uint32_t cap, uint32_t n, uint32_t chunk, bool& overflow,
                                              Hole hole) {
        uint32_t total;
        const uint32_t pre = waveExclusiveScan(n, total);
        total = __builtin_amdgcn_readfirstlane(total);
        if (total == 0) return 0xFFFFFFFFu;
        if (used + total > size) {
            fill(hole);
            const uint32_t want = total > chunk ? total : chunk;
            uint32_t b = 0;
            if ((threadIdx.x & 63u) == 0) b = atomicAdd(counter, want);
            b = __builtin_amdgcn_readfirstlane(__shfl(b, 0)); // (wave-uniform: kept in a scalar register)
            if (b > cap || want > cap - b) { // (the counter keeps the needed size for the retry on the host)
                overflow = true;
                size = used = 0;
                return 0xFFFFFFFFu;
            }
            base = b;
            size = want;
            used = 0;
        }
        const uint32_t o = base + used + pre;
        used += total;
        return o;
    }
    // the same with the prefix sum done by the caller: `pre` = the records of the lanes below, `total` (wave-uniform) of all lanes
    template <class Hole>
    __device__ __forceinline__ uint32_t allocPre(uint32_t* counter, uint32_t cap, uint32_t pre, uint32_t total, uint32_t chunk, bool& overflow,
                                                 Hole hole) {
        if (total == 0) return 0xFFFFFFFFu;
        if (used + total > size) {
            fill(hole);
            const uint32_t want = total > chunk ? total : chunk;
            uint32_t b = 0;
            if ((threadIdx.x & 63u) == 0) b = atomicAdd(counter, want);
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            if (b > cap || want > cap - b) {
                overflow = true;
                size = used = 0;
                return 0xFFFFFFFFu;
            }
            base = b;
            size = want;
            used = 0;
        }
        const uint32_t o = base + used + pre;
        used += total;
        return o;
    }
    template <class Hole>
    __device__ __forceinline__ void fill(Hole hole) { // holes in the unused rest of the current chunk
        for (uint32_t i = used + (threadIdx.x & 63u); i < size; i += 64u) hole(base + i);
        used = size;
    }
};

} // namespace cmb

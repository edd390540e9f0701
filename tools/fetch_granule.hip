// What does FETCH_SIZE count for SCATTERED loads on gfx950?  ("weak" 4)
//
// One kernel per access shape, each over the same N random 128-byte lines of a 4 GiB table (far beyond the 256 MiB
// Infinity Cache and the L2): per line the lane requests
//     k_touch<1>   one 16-byte chunk                       (16 B requested)
//     k_touch<2>   the first 32-byte sector                (32 B: one rank block of dev_index.hpp)
//     k_touch<4>   the first 64-byte half                  (64 B: what a 192-position rank block asked of its line)
//     k_touch<8>   all 128 bytes                           (128 B)
// and, for scale, k_stream reads the same number of bytes as k_touch<8> as a fully coalesced 16-B-per-lane stream (the
// case MI355X_MICROARCH.md calibrates: FETCH_SIZE = half the bytes).
// Run under `rocprofv3 --pmc FETCH_SIZE` (tools/calibrate_fetch.sh); the program prints the known bytes per line, the
// script divides the counter by N.  If raw FETCH per line goes 64 -> 64 -> 64 -> 128 the counter is exact at a 64-byte
// granule for scattered loads; if it stays at 64 for all four, a whole 128-byte line is fetched and tallied at half.
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/fetch_granule tools/fetch_granule.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x)                                                                                  \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));          \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

template <int CHUNKS>
__global__ void __launch_bounds__(256) k_touch(const uint4* __restrict__ table, const uint32_t* __restrict__ lines, uint32_t n,
                                               uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint4* L = table + (size_t)lines[i] * 8;
        uint4 v[CHUNKS];
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) v[c] = L[c];
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) acc ^= v[c].x ^ v[c].y ^ v[c].z ^ v[c].w;
    }
    if (acc == 0x12345678u) sink[0] = acc; // (keeps the loads alive; practically never true)
}

__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ table, uint64_t nChunks, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nChunks; i += (uint64_t)gridDim.x * 256ull) {
        const uint4 v = table[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const uint32_t N = 1u << 24;
    const uint64_t tableBytes = 4ull << 30, nLines = tableBytes / 128;
    uint4* table;
    uint32_t *lines, *sink;
    CHK(hipMalloc(&table, tableBytes));
    CHK(hipMalloc(&lines, (size_t)N * 4));
    CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(table, 1, tableBytes));
    // distinct random lines: a fixed odd multiplier modulo 2^25 lines is a permutation
    uint32_t* h = (uint32_t*)malloc((size_t)N * 4);
    for (uint32_t i = 0; i < N; i++) h[i] = (uint32_t)(((uint64_t)i * 2654435761ull + 12345ull) % nLines);
    CHK(hipMemcpy(lines, h, (size_t)N * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    const int reps = 5;
    auto timeit = [&](const char* name, auto launch, double bytesRequested) {
        launch();
        CHK(hipEventRecord(a, 0));
        for (int r = 0; r < reps; r++) launch();
        CHK(hipEventRecord(b, 0));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        ms /= reps;
        printf("%-10s %8.3f ms  requested %6.1f B per line  %7.1f GB/s requested  %6.2f G lines/s\n", name, ms, bytesRequested,
               bytesRequested * N / ms * 1e-6, N / ms * 1e-6);
    };
    const unsigned grid = 256 * 16;
    timeit("touch16", [&] { hipLaunchKernelGGL(k_touch<1>, dim3(grid), dim3(256), 0, 0, table, lines, N, sink); }, 16);
    timeit("touch32", [&] { hipLaunchKernelGGL(k_touch<2>, dim3(grid), dim3(256), 0, 0, table, lines, N, sink); }, 32);
    timeit("touch64", [&] { hipLaunchKernelGGL(k_touch<4>, dim3(grid), dim3(256), 0, 0, table, lines, N, sink); }, 64);
    timeit("touch128", [&] { hipLaunchKernelGGL(k_touch<8>, dim3(grid), dim3(256), 0, 0, table, lines, N, sink); }, 128);
    timeit("stream", [&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, table, (uint64_t)N * 8, sink); }, 128);
    CHK(hipDeviceSynchronize());
    printf("N = %u lines per launch, %d + 1 launches per shape, lines[] itself: 4 B per line (coalesced)\n", N, reps);
    return 0;
}

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "dev_wave.hpp"
__global__ void k(const uint32_t* in, uint32_t* out) {
    const uint32_t v = in[threadIdx.x];
    const uint32_t x = cmb::waveInclusiveScanDpp(v);
    out[threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) out[256 + (threadIdx.x >> 6)] = cmb::waveLastLane(x);
}
int main() {
    uint32_t *in, *out; hipMalloc(&in, 1024); hipMalloc(&out, 2048);
    uint32_t h[256], o[260]; int bad = 0;
    for (int t = 0; t < 20; t++) {
        for (int i = 0; i < 256; i++) h[i] = (uint32_t)((i * 2654435761u + t * 40503u) >> 27) % (t < 10 ? 5 : 300);
        hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
        k<<<1, 256>>>(in, out);
        hipMemcpy(o, out, 260 * 4, hipMemcpyDeviceToHost);
        for (int w = 0; w < 4; w++) { uint32_t s = 0; for (int l = 0; l < 64; l++) { s += h[w * 64 + l]; if (o[w * 64 + l] != s) { if (bad < 5) printf("t %d w %d l %d got %u want %u\n", t, w, l, o[w*64+l], s); bad++; } } if (o[256 + w] != s) bad++; }
    }
    printf("dpp scan test: %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ void gldsU4(const uint4* src, uint4* ldsWaveBase) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_ptr_t)ldsWaveBase);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
__global__ void k(const uint4* __restrict__ src, uint4* __restrict__ out, uint32_t mask_lo) {
    __shared__ uint4 pad[7];
    __shared__ uint4 lds[2][256];
    const uint32_t tid = threadIdx.x;
    if (tid < 7) pad[tid] = make_uint4(1, 2, 3, 4);
    lds[0][tid] = make_uint4(0xAAAAAAAAu, 0, 0, tid);
    lds[1][tid] = make_uint4(0xBBBBBBBBu, 0, 0, tid);
    __syncthreads();
    const bool act = ((tid * 2654435761u) >> 7) & 1u ? true : ((tid & 63u) >= mask_lo);
    if (act) {
        const uint4* s = src + (tid * 7u) % 1024u;
        gldsU4(s, &lds[0][tid & ~63u]);
        gldsU4(s + 1024, &lds[1][tid & ~63u]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[tid] = lds[0][tid];
    out[256 + tid] = lds[1][tid];
    if (tid < 7) out[512 + tid] = pad[tid];
}
int main() {
    uint4 *src, *out;
    hipMalloc(&src, 2048 * 16); hipMalloc(&out, 520 * 16);
    uint4 h[2048];
    for (int i = 0; i < 2048; i++) h[i] = make_uint4(i, i * 3, i * 5, i * 7);
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    int bad = 0;
    for (uint32_t mlo : {0u, 1u, 17u, 40u}) {
        k<<<1, 256>>>(src, out, mlo);
        uint4 o[520];
        hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
        for (uint32_t t = 0; t < 256; t++) {
            const bool act = ((t * 2654435761u) >> 7) & 1u ? true : ((t & 63u) >= mlo);
            uint32_t e = (t * 7u) % 1024u;
            uint32_t w0 = act ? e : 0xAAAAAAAAu, w1 = act ? e + 1024 : 0xBBBBBBBBu;
            if (o[t].x != w0 || o[256 + t].x != w1) { if (bad < 10) printf("mlo %u lane %u act %d got %x %x want %x %x\n", mlo, t, act, o[t].x, o[256+t].x, w0, w1); bad++; }
        }
        for (int t = 0; t < 7; t++) if (o[512 + t].x != 1 || o[512+t].w != 4) bad++;
    }
    printf("asm glds test: %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}

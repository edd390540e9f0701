// Issue rate of the integer vector instructions the search kernels are made of (MI355X, gfx950): cycles per wave64 instruction per SIMD
// with W wavefronts per SIMD, each running a long stream of INDEPENDENT instructions of one kind.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(x) x x x x x x x x
#define OPS(name, body)                                                                                   \
    __global__ void __launch_bounds__(256) k_##name(uint32_t* out, uint32_t n, uint32_t seed) {          \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u; \
        uint32_t b = seed * 31u + 7u;                                                                      \
        uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;                                                       \
        for (uint32_t i = 0; i < n; i++) { REP8(body) }                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3); \
    }
// eight independent instructions per body
OPS(add, asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
OPS(and, asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
OPS(bitop3, asm volatile("v_bitop3_b32 %0, %0, %8, %1 bitop3:0xde\n v_bitop3_b32 %1, %1, %8, %2 bitop3:0xde\n v_bitop3_b32 %2, %2, %8, %3 bitop3:0xde\n v_bitop3_b32 %3, %3, %8, %4 bitop3:0xde\n v_bitop3_b32 %4, %4, %8, %5 bitop3:0xde\n v_bitop3_b32 %5, %5, %8, %6 bitop3:0xde\n v_bitop3_b32 %6, %6, %8, %7 bitop3:0xde\n v_bitop3_b32 %7, %7, %8, %0 bitop3:0xde" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
OPS(shl, asm volatile("v_lshlrev_b32 %0, %8, %0\n v_lshlrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3\n v_lshlrev_b32 %4, %8, %4\n v_lshlrev_b32 %5, %8, %5\n v_lshlrev_b32 %6, %8, %6\n v_lshlrev_b32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
OPS(bcnt, asm volatile("v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
OPS(cndmask, asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
OPS(cndmask64, asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[22:23]\n v_cndmask_b32_e64 %5, %5, %8, s[22:23]\n v_cndmask_b32_e64 %6, %6, %8, s[22:23]\n v_cndmask_b32_e64 %7, %7, %8, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "s23");)
OPS(cmp64, asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %8\n v_cmp_lt_u32_e64 s[22:23], %1, %8\n v_cmp_lt_u32_e64 s[24:25], %2, %8\n v_cmp_lt_u32_e64 s[26:27], %3, %8\n v_cmp_lt_u32_e64 s[20:21], %4, %8\n v_cmp_lt_u32_e64 s[22:23], %5, %8\n v_cmp_lt_u32_e64 s[24:25], %6, %8\n v_cmp_lt_u32_e64 s[26:27], %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
OPS(cmpcnd, asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %8\n v_add_u32 %4, %4, %8\n v_cmp_lt_u32_e64 s[22:23], %1, %8\n v_add_u32 %5, %5, %8\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_add_u32 %6, %6, %8\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "s23");)
OPS(add3, asm volatile("v_add3_u32 %0, %0, %8, %1\n v_add3_u32 %1, %1, %8, %2\n v_add3_u32 %2, %2, %8, %3\n v_add3_u32 %3, %3, %8, %4\n v_add3_u32 %4, %4, %8, %5\n v_add3_u32 %5, %5, %8, %6\n v_add3_u32 %6, %6, %8, %7\n v_add3_u32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
OPS(cmp, asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
OPS(fma, asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
OPS(shl64, asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3\n v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b));)

typedef void (*kern_t)(uint32_t*, uint32_t, uint32_t);
int main() {
    uint32_t* out;
    hipMalloc(&out, 256 * 8192 * 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    const int cus = p.multiProcessorCount;
    struct { const char* n; kern_t k; int perBody; } ks[] = {{"v_add_u32", k_add, 8}, {"v_and_b32", k_and, 8}, {"v_bitop3_b32", k_bitop3, 8}, {"v_lshlrev_b32", k_shl, 8},
        {"v_bcnt_u32_b32", k_bcnt, 8}, {"v_cndmask_b32", k_cndmask, 8}, {"v_cmp_lt_u32", k_cmp, 8}, {"v_cndmask_e64 sgpr", k_cndmask64, 8}, {"v_cmp_e64 -> sgpr", k_cmp64, 8}, {"cmp,add,cmp,add,cnd,add,cnd,add", k_cmpcnd, 8}, {"v_add3_u32", k_add3, 8}, {"v_fma_f32", k_fma, 8}, {"v_lshlrev_b64", k_shl64, 8}};
    printf("%d CUs at %.2f GHz (clockRate); cycles per wave64 instruction per SIMD\n", cus, ghz);
    for (auto& e : ks) {
        printf("%-28s", e.n);
        for (int wps : {1, 2, 4, 8}) { // wavefronts per SIMD: blocks of 256 threads = one wave per SIMD of a CU
            const uint32_t n = 4096;
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            e.k<<<cus * wps, 256>>>(out, 64, 1); // warm-up
            hipEventRecord(a);
            e.k<<<cus * wps, 256>>>(out, n, 1);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double instPerSimd = (double)n * 8 * e.perBody * wps;
            printf("  W=%d: %5.2f", wps, ms * 1e-3 * ghz * 1e9 / instPerSimd);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

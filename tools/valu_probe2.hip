// tools/valu_probe2.hip -- issue cost of specific encodings (inline asm, 8 independent registers, 4 waves per SIMD):
// is a VOP3-encoded instruction slower than the same operation in VOP2 encoding, and which of the instructions the
// walker is made of are the expensive ones?  Measurement tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define BODY(NAME, ASM)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t seed)                  \
    {                                                                                                     \
        uint32_t r[8];                                                                                    \
        for (int k = 0; k < 8; k++) r[k] = seed * (threadIdx.x + 1) + k * 77u;                            \
        uint32_t a = seed ^ 0x55u, b = threadIdx.x;                                                       \
        for (int i = 0; i < iters; i++) {                                                                 \
            _Pragma("unroll") for (int u = 0; u < 4; u++) { _Pragma("unroll") for (int k = 0; k < 8; k++) { ASM; } } \
        }                                                                                                 \
        uint32_t s = 0;                                                                                   \
        for (int k = 0; k < 8; k++) s ^= r[k];                                                            \
        if (s == 0x12345678u) out[0] = s;                                                                 \
    }

BODY(k_and_e32, asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(r[k]) : "v"(a)))
BODY(k_and_e64, asm volatile("v_and_b32_e64 %0, %1, %0" : "+v"(r[k]) : "v"(a)))
BODY(k_and_e64_sgpr, asm volatile("v_and_b32_e64 %0, %0, %1" : "+v"(r[k]) : "s"(seed)))
BODY(k_cndmask_e32, asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[k]) : "v"(a) : "vcc"))
BODY(k_cndmask_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(r[k]) : "v"(a) : "s10", "s11"))
BODY(k_cmp_e32, asm volatile("v_cmp_eq_u32_e32 vcc, %1, %0" : "+v"(r[k]) : "v"(a) : "vcc"))
BODY(k_cmp_e64, asm volatile("v_cmp_eq_u32_e64 s[10:11], %1, %0" : "+v"(r[k]) : "v"(a) : "s10", "s11"))
BODY(k_add3, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b)))
BODY(k_perm, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b)))
BODY(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(r[k]) : "v"(a)))
BODY(k_alignbyte_v, asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b)))
BODY(k_lshlrev_b64, asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(*reinterpret_cast<uint64_t *>(&r[k & 6]))))
BODY(k_mov, asm volatile("v_mov_b32_e32 %0, %1" : "+v"(r[k]) : "v"(a)))
BODY(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[k]) : "v"(a)))
BODY(k_mul_u24, asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(r[k]) : "v"(a)))
BODY(k_bfe, asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(r[k])))
BODY(k_lshrrev, asm volatile("v_lshrrev_b32_e32 %0, 3, %0" : "+v"(r[k])))
BODY(k_lshrrev_sgpr, asm volatile("v_lshrrev_b32_e32 %0, %1, %0" : "+v"(r[k]) : "s"(seed)))
BODY(k_lshrrev_vgpr, asm volatile("v_lshrrev_b32_e32 %0, %1, %0" : "+v"(r[k]) : "v"(a)))
BODY(k_and_lit, asm volatile("v_and_b32_e32 %0, 0xfffc, %0" : "+v"(r[k])))
BODY(k_and_inl, asm volatile("v_and_b32_e32 %0, -4, %0" : "+v"(r[k])))
BODY(k_mul_u24_vgpr, asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(r[k]) : "v"(a)))
BODY(k_mul_u24_lit, asm volatile("v_mul_u32_u24_e32 %0, 0x8b92c5, %0" : "+v"(r[k])))
BODY(k_add_e32, asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(r[k]) : "v"(a)))
BODY(k_lshl_or, asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r[k]) : "v"(a)))
BODY(k_ds_read, asm volatile("ds_read_b32 %0, %1" : "+v"(r[k]) : "v"(b)))
BODY(k_and_or, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b)))
BODY(k_salu, asm volatile("s_add_u32 s10, s10, 1" ::: "s10", "scc"))
BODY(k_mbcnt, asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r[k]) : "v"(a)))

typedef void (*kern)(uint32_t *, int, uint32_t);
static void run(const char *name, kern f, uint32_t *d, int cus)
{
    const int iters = 20000, occ = 4;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(f, dim3(cus * occ), dim3(256), 0, 0, d, 100, 1u);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(f, dim3(cus * occ), dim3(256), 0, 0, d, iters, 3u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-22s %.3f ns per wave-instruction per SIMD (4 waves/SIMD)\n", name, ms * 1e6 / ((double)occ * iters * 32.0));
}

int main()
{
    uint32_t *d; (void)hipMalloc(&d, 4);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
#define R(n) run(#n, n, d, cus)
    R(k_mov); R(k_and_e32); R(k_and_e64); R(k_and_e64_sgpr); R(k_lshrrev); R(k_mul_u24); R(k_mul_lo);
    R(k_cndmask_e32); R(k_cndmask_e64); R(k_cmp_e32); R(k_cmp_e64); R(k_add3); R(k_and_or); R(k_perm);
    R(k_lshrrev_sgpr); R(k_lshrrev_vgpr); R(k_and_lit); R(k_and_inl); R(k_mul_u24_vgpr); R(k_mul_u24_lit); R(k_add_e32); R(k_lshl_or);
    R(k_alignbit); R(k_alignbyte_v); R(k_bfe); R(k_lshlrev_b64); R(k_mbcnt); R(k_salu);
    return 0;
}

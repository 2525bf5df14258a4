// tools/valu_probe.hip -- issue rate of the integer VALU instructions the level-1 filter is made of, relative
// to v_fma_f32 (2 cycles per wave64 instruction on a SIMD-32, MI355X_MICROARCH.md).  One block of 256 threads
// per CU x OCC, every thread runs a long chain of 8 independent instruction streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP> __device__ __forceinline__ uint32_t op(uint32_t a, uint32_t b)
{
    if (OP == 0) return __float_as_uint(__builtin_fmaf(__uint_as_float(a), 1.0001f, __uint_as_float(b)));
    if (OP == 1) return (a & b) + 1u;                                   // v_and + v_add  (2 instr)
    if (OP == 2) return __builtin_amdgcn_alignbyte(a, b, 1);
    if (OP == 3) return (uint32_t)__umul24(a, 0x797A0Bu);
    if (OP == 4) return __builtin_amdgcn_ubfe(a, b, 3u);
    if (OP == 5) return (a << 3) | b;                                   // v_lshl_or_b32
    if (OP == 6) return a * 0x9E3779B1u;                                // v_mul_lo_u32
    if (OP == 7) return __umulhi(a & 0xFFFFFFu, 0x797A0Bu);             // v_and + v_mul_hi_u32 (or u24)
    if (OP == 8) return (uint32_t)__builtin_popcount(a) + b;            // v_bcnt_u32_b32 (has an add built in)
    return a ^ b;
}

template <int OP> __global__ __launch_bounds__(256) void probe(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t r[8];
    for (int k = 0; k < 8; k++) r[k] = seed * (threadIdx.x + 1) + k * 77u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int k = 0; k < 8; k++) r[k] = op<OP>(r[k], r[(k + 1) & 7]);
    }
    uint32_t s = 0;
    for (int k = 0; k < 8; k++) s ^= r[k];
    if (s == 0x12345678u) out[0] = s;
}

template <int OP> void run(const char *name, int instrPerOp, uint32_t *d, int cus)
{
    for (int occ : {1, 2, 4}) {
        const int iters = 20000;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        probe<OP><<<cus * occ, 256>>>(d, 100, 1u);
        hipEventRecord(a);
        probe<OP><<<cus * occ, 256>>>(d, iters, 3u);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // per SIMD: occ waves (256 threads = 4 waves = 1 per SIMD per block) x iters x 32 ops
        const double opsPerSimd = (double)occ * iters * 32.0 * instrPerOp;
        printf("%-28s occ %d waves/SIMD: %.3f ns per wave-instruction per SIMD (%.2f ms)\n", name, occ, ms * 1e6 / opsPerSimd, ms);
    }
}

int main()
{
    uint32_t *d; hipMalloc(&d, 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("CUs %d clock %d kHz\n", cus, p.clockRate);
    run<0>("v_fma_f32", 1, d, cus);
    run<1>("v_and+v_add", 2, d, cus);
    run<2>("v_alignbyte", 1, d, cus);
    run<3>("v_mul_u32_u24", 1, d, cus);
    run<4>("v_bfe_u32", 1, d, cus);
    run<5>("v_lshl_or_b32", 1, d, cus);
    run<6>("v_mul_lo_u32", 1, d, cus);
    run<7>("v_and+v_mul_hi", 2, d, cus);
    run<8>("v_bcnt", 1, d, cus);
    return 0;
}

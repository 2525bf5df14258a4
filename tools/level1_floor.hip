// tools/level1_floor.hip -- what LEVEL 1 of the compacted-output kernel costs with nothing behind it: the floor of any design that asks one
// LDS bit per input position (measurement tool, not product; `hipcc -O3 --offload-arch=gfx950 -o /tmp/level1_floor tools/level1_floor.hip`).
//
// One persistent 1024-thread block per CU like pfac_scan_filter<REDUCE>; every wave streams 1 KiB tiles of a 1 GiB input with 16-byte loads and
// tests its 16 positions per lane and tile exactly as the kernel does (scan_filter.hip, step 1, REDUCE: gram -- a shift or v_alignbyte --,
// v_mul_u32_u24, ONE SDWA AND for the byte address, ds_read_b32 from a 64 KiB bitmap, a shift by the gram, v_alignbit into the lane's mask:
// five vector instructions and one LDS read per position), and counts the hits.  Three address modes:
//   hashed      the kernel's: the product's high half picks the dword -- 64 lanes at 64 random places of the 32 LDS banks
//   lane        every lane reads its own bank (address = lane * 4 + a row the hash picks per WAVE): no bank conflict by construction.  NOT a
//               filter anybody can use -- a 3-gram's bit would have to exist in every bank, i.e. a bitmap of 1/32 of the size -- it prices the conflicts
//   none        no LDS read at all (the word is the product itself): the five vector instructions alone
// Prints ms per GiB and the implied rate.  Round 6 on MI355X (profiles/r06_level1_floor.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kMul = 0x8B92C5u;            // pfac::kGram1Mul
constexpr uint32_t kTableBytes = 64 * 1024;     // gram1: 2^19 bits

template <int MODE>
__global__ __launch_bounds__(1024) void level1(const u32x4 *in, size_t tiles, const uint32_t *bitmap, unsigned long long *hitsOut, unsigned int *next)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < kTableBytes / 4; i += 1024) lds[i] = bitmap[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t vMask, vMul;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vMask) : "s"(0xFFFCu));
    asm volatile("v_mov_b32 %0, %1" : "=v"(vMul) : "s"(kMul));
    unsigned long long found = 0;
    for (;;) {
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(next, 64u);                    // 64 tiles per claim, in order (one device counter answers ~90 atomics per microsecond: 16 K claims per GiB)
        t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= tiles) break;
        // four tiles in flight per wave (the kernel's chunk prefetch is 2 KiB ahead): the loads and their waits are written out so that a wait
        // leaves the three younger loads in flight (the compiler's own wait would be for all of them).  A claim is 64 whole tiles (2^20 tiles).
        u32x4 d0, d1, d2, d3;
        auto fetch = [&](u32x4 &d, unsigned int tileNo) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(in + (size_t)tileNo * 64 + lane) : "memory"); };
        fetch(d0, t); fetch(d1, t + 1); fetch(d2, t + 2); fetch(d3, t + 3);
        auto tile = [&](const u32x4 d) {
            const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
            uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dw[0], 0x130, 0xf, 0xf, false);   // the next lane's first dword (the tile's last lane: 0)
            uint32_t hits = 0;
#pragma unroll
            for (int b0 = 0; b0 < 16; b0 += 8) {
                uint32_t word[8], xs[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int j = (b0 + q) >> 2, i = (b0 + q) & 3;
                    const uint32_t nx = j < 3 ? dw[(j + 1) & 3] : nxt;
                    const uint32_t x = i == 0 ? dw[j] : i == 1 ? dw[j] >> 8 : __builtin_amdgcn_alignbyte(nx, dw[j], i);
                    const uint32_t product = (uint32_t)__umul24(x, vMul);
                    uint32_t addr;
                    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(addr) : "v"(product), "v"(vMask));
                    if (MODE == 0) word[q] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(addr);
                    else if (MODE == 1) word[q] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(((uint32_t)__builtin_amdgcn_readfirstlane((int)addr) & 0xFF00u) + lane * 4u);
                    else word[q] = addr;
                    xs[q] = x;
                }
#pragma unroll
                for (int q = 0; q < 8; q++) hits = __builtin_amdgcn_alignbit(word[q] >> (xs[q] & 31u), hits, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            found += (unsigned long long)__builtin_popcount(hits);
        };
        for (unsigned int k = 0; k < 64; k += 4) {
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(d0)); tile(d0); fetch(d0, t + k + 4);
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(d1)); tile(d1); fetch(d1, t + k + 5);
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(d2)); tile(d2); fetch(d2, t + k + 6);
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(d3)); tile(d3); fetch(d3, t + k + 7);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
    }
    if (found == 0xFFFFFFFFFFFFull) *hitsOut = found;               // keeps the work
    atomicAdd(hitsOut + 1, found);
}

template <int MODE>
static double run(const char *name, const u32x4 *d_in, size_t n, const uint32_t *d_map, int cus)
{
    unsigned long long *d_hits;
    unsigned int *d_next;
    (void)hipMalloc(&d_hits, 16);
    (void)hipMalloc(&d_next, 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(level1<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTableBytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e9f;
    unsigned long long hits[2] = {0, 0};
    for (int r = 0; r < 12; r++) {
        (void)hipMemset(d_hits, 0, 16);
        (void)hipMemset(d_next, 0, 4);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(level1<MODE>, dim3(cus), dim3(1024), kTableBytes, 0, d_in, n / 1024, d_map, d_hits, d_next);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (r >= 2 && ms < best) best = ms;
    }
    (void)hipMemcpy(hits, d_hits, 16, hipMemcpyDeviceToHost);
    printf("%-8s %.4f ms per GiB = %.0f GB/s of input, %.2f %% of the positions pass\n", name, best * (double)(1 << 30) / (double)n, (double)n / best / 1e6,
           100.0 * (double)hits[1] / (double)n);
    (void)hipFree(d_hits);
    (void)hipFree(d_next);
    return best;
}

int main()
{
    const size_t n = size_t(1) << 30;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    std::vector<uint32_t> text(n / 4);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    const char alpha[] = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-._~/?=&%+ :\r\n";
    for (size_t i = 0; i < n / 4; i++) {
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) { s = s * 6364136223846793005ull + 1442695040888963407ull; w |= (uint32_t)(unsigned char)alpha[(s >> 33) % (sizeof(alpha) - 1)] << (8 * k); }
        text[i] = w;
    }
    std::vector<uint32_t> map(kTableBytes / 4, 0);
    for (int i = 0; i < 25000; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; map[(s >> 40) % map.size()] |= 1u << ((s >> 20) & 31); }   // 25 000 3-grams: 4.8 % of 2^19 bits
    u32x4 *d_in;
    uint32_t *d_map;
    (void)hipMalloc(&d_in, n + 65536);
    (void)hipMalloc(&d_map, kTableBytes);
    (void)hipMemcpy(d_in, text.data(), n, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_map, map.data(), kTableBytes, hipMemcpyHostToDevice);
    printf("level 1 of the compacted-output kernel alone, %d CUs x 16 waves, 1 GiB of text, one LDS bit per position:\n", p.multiProcessorCount);
    run<0>("hashed", d_in, n, d_map, p.multiProcessorCount);
    run<1>("lane", d_in, n, d_map, p.multiProcessorCount);
    run<2>("none", d_in, n, d_map, p.multiProcessorCount);
    return 0;
}

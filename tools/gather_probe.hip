// tools/gather_probe.hip -- what does a divergent gather cost on MI355X?  (dev tool, GPU box only)
//   hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
// Every lane runs U independent dependent-load chains over a random table (the walkers' access
// pattern); reports lane-loads per cycle per CU for 4/8/16/2x16 byte elements and table sizes that
// live in L1 / L2 / MALL.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));

template <int BYTES, int U, bool UNALIGNED>
__global__ __launch_bounds__(1024) void gather(const unsigned char *table, uint32_t mask, int rounds, uint32_t *sink)
{
    uint32_t st[U];
    for (int u = 0; u < U; u++) st[u] = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + u * 0x9E3779B9u;
    for (int r = 0; r < rounds; r++) {
        uint32_t got[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t idx = (st[u] * 0x85EBCA6Bu) >> 7;
            idx &= mask;                                   // element index
            const unsigned char *p = table + (size_t)idx * 32u + (UNALIGNED ? (st[u] & 15u) : 0u);
            if (BYTES == 4) got[u] = *reinterpret_cast<const uint32_t *>(p);
            else if (BYTES == 8) { u32x2 v = *reinterpret_cast<const u32x2 *>(p); got[u] = v.x ^ v.y; }
            else if (BYTES == 16) {
                u32x4 v = UNALIGNED ? u32x4(*reinterpret_cast<const u32x4_a1 *>(p)) : *reinterpret_cast<const u32x4 *>(p);
                got[u] = v.x ^ v.y ^ v.z ^ v.w;
            } else {
                u32x4 v = *reinterpret_cast<const u32x4 *>(p), w = *reinterpret_cast<const u32x4 *>(p + 16);
                got[u] = v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) st[u] = st[u] * 1664525u + got[u] + 1013904223u;
    }
    uint32_t acc = 0;
    for (int u = 0; u < U; u++) acc ^= st[u];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int BYTES, int U, bool UNALIGNED>
void run(const char *name, const unsigned char *table, size_t tableBytes, uint32_t *sink, int cus, double mhz)
{
    const uint32_t mask = (uint32_t)(tableBytes / 32 - 1);
    const int rounds = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    gather<BYTES, U, UNALIGNED><<<cus, 1024>>>(table, mask, 100, sink);
    hipEventRecord(a);
    gather<BYTES, U, UNALIGNED><<<cus, 1024>>>(table, mask, rounds, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double laneLoads = (double)cus * 1024 * U * rounds * (BYTES == 32 ? 2 : 1);
    const double perCuPerUs = laneLoads / cus / (ms * 1e3);
    printf("%-28s table %7.2f MB  U=%d  %8.3f ms  round %6.2f us  %7.1f lane-loads/us/CU  (%.2f cycles per lane-load at %.0f MHz)\n",
           name, tableBytes / 1e6, U, ms, ms * 1e3 / rounds, perCuPerUs, mhz / perCuPerUs, mhz);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1e3;
    printf("%s: %d CUs, %.0f MHz\n", prop.gcnArchName, cus, mhz);
    const size_t maxBytes = 32u << 20;
    std::vector<uint32_t> h(maxBytes / 4);
    uint64_t x = 88172645463325252ull;
    for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)x; }
    unsigned char *d; uint32_t *sink;
    hipMalloc(&d, maxBytes + 64); hipMalloc(&sink, 64);
    hipMemcpy(d, h.data(), maxBytes, hipMemcpyHostToDevice);
    for (size_t tb : {size_t(8) << 10, size_t(2) << 20, size_t(32) << 20}) {
        run<4, 3, false>("dword", d, tb, sink, cus, mhz);
        run<8, 3, false>("dwordx2", d, tb, sink, cus, mhz);
        run<16, 3, false>("dwordx4", d, tb, sink, cus, mhz);
        run<16, 3, true>("dwordx4 unaligned", d, tb, sink, cus, mhz);
        run<32, 3, false>("2 x dwordx4 (32 B slot)", d, tb, sink, cus, mhz);
        run<16, 1, false>("dwordx4", d, tb, sink, cus, mhz);
        run<16, 6, false>("dwordx4", d, tb, sink, cus, mhz);
    }
    return 0;
}

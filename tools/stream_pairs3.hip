// tools/stream_pairs3.hip -- follow-up: the wave that reads input tile t writes the zeros of tile (t + lead) mod T.
// For a slow and a fast pair of allocations (1 GiB / 4 GiB, hipMalloc), sweep the lead: does the class of a pair depend
// on which part of the result buffer is written WHILE a given part of the input is read?  Measurement tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void r1w4(const u32x4 *in, i32x4 *out, unsigned *sink, unsigned lead, unsigned tiles)
{
    const unsigned tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned wt = (tile + lead) & (tiles - 1);
    const int lane = threadIdx.x & 63;
    const u32x4 v = in[(size_t)tile * 64 + lane];
    const i32x4 z = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) __builtin_nontemporal_store(z, &out[(size_t)wt * 256 + k * 64 + lane]);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = v.x;
}
static const size_t N = size_t(1) << 30;
static unsigned *sink; static hipEvent_t ea, eb;
static float timeit(const void *in, void *out, unsigned lead)
{
    const unsigned tiles = N / 1024;
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(r1w4, dim3(tiles / 4), dim3(256), 0, 0, (const u32x4 *)in, (i32x4 *)out, sink, lead, tiles);
    (void)hipEventRecord(ea);
    for (int r = 0; r < 8; r++) hipLaunchKernelGGL(r1w4, dim3(tiles / 4), dim3(256), 0, 0, (const u32x4 *)in, (i32x4 *)out, sink, lead, tiles);
    (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
    float ms; (void)hipEventElapsedTime(&ms, ea, eb); return ms / 8;
}
int main()
{
    (void)hipMalloc(&sink, 4); (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    char *in[4], *out[4];
    for (int k = 0; k < 4; k++) { (void)hipMalloc(&in[k], N); (void)hipMalloc(&out[k], 4 * N); (void)hipMemset(in[k], k + 1, N); }
    int si = 0, sj = 0, fi = 0, fj = 0; float smax = 0, fmin = 1e9;
    for (int i = 0; i < 4; i++) { printf("in %d:", i); for (int j = 0; j < 4; j++) { float t = timeit(in[i], out[j], 0); printf(" %.4f", t); if (t > smax) { smax = t; si = i; sj = j; } if (t < fmin) { fmin = t; fi = i; fj = j; } } printf("\n"); }
    printf("slowest pair in%d/out%d %.4f, fastest in%d/out%d %.4f\n", si, sj, smax, fi, fj, fmin);
    printf("%14s %12s %12s\n", "lead (tiles)", "slow pair", "fast pair");
    for (unsigned lead : {0u, 1u, 2u, 4u, 8u, 16u, 32u, 64u, 128u, 256u, 512u, 1024u, 2048u, 4096u, 8192u, 16384u, 32768u, 65536u, 131072u, 262144u, 524288u, 3u, 5u, 96u, 1536u, 49152u, 393216u})
        printf("%14u %12.4f %12.4f\n", lead, timeit(in[si], out[sj], lead), timeit(in[fi], out[fj], lead));
    return 0;
}

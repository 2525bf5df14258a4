// tools/stream_pairs2.hip -- follow-up to stream_pairs.hip: for a slow and a fast (input, result) pair, shift the result
// pointer (and, separately, the input pointer) inside its allocation by d = 4 KiB ... 1 GiB: does the class of the pair
// depend on the relative position of the two streams?  Measurement tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void r1w4(const u32x4 *in, i32x4 *out, unsigned *sink)
{
    const size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const u32x4 v = in[tile * 64 + lane];
    const i32x4 z = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) __builtin_nontemporal_store(z, &out[tile * 256 + k * 64 + lane]);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = v.x;
}
static const size_t N = size_t(1) << 30;
static unsigned *sink; static hipEvent_t ea, eb;
static float timeit(const void *in, void *out)
{
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(r1w4, dim3(N / 4096), dim3(256), 0, 0, (const u32x4 *)in, (i32x4 *)out, sink);
    (void)hipEventRecord(ea);
    for (int r = 0; r < 8; r++) hipLaunchKernelGGL(r1w4, dim3(N / 4096), dim3(256), 0, 0, (const u32x4 *)in, (i32x4 *)out, sink);
    (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
    float ms; (void)hipEventElapsedTime(&ms, ea, eb); return ms / 8;
}
int main()
{
    (void)hipMalloc(&sink, 4); (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    const size_t pad = size_t(2) << 30;
    char *in[3], *out[3];
    for (int k = 0; k < 3; k++) { (void)hipMalloc(&in[k], N + pad); (void)hipMalloc(&out[k], 4 * N + pad); (void)hipMemset(in[k], k + 1, N + pad); }
    printf("in : %p %p %p\nout: %p %p %p\n", in[0], in[1], in[2], out[0], out[1], out[2]);
    int si = 0, sj = 0, fi = 0, fj = 0; float smax = 0, fmin = 1e9;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float t = timeit(in[i], out[j]); printf("  pair in%d/out%d %.4f\n", i, j, t); if (t > smax) { smax = t; si = i; sj = j; } if (t < fmin) { fmin = t; fi = i; fj = j; } }
    printf("slowest pair in%d/out%d %.4f, fastest in%d/out%d %.4f\n", si, sj, smax, fi, fj, fmin);
    const size_t offs[] = {0, 1u << 12, 1u << 13, 1u << 14, 1u << 15, 1u << 16, 1u << 17, 1u << 18, 1u << 19, 1u << 20, 1u << 21, 1u << 22, 1u << 23, 1u << 24, 1u << 25, 1u << 26, 1u << 27, 1u << 28, 1u << 29, 1u << 30, 3u << 20, 5u << 22, 7u << 24};
    printf("%12s %22s %22s %22s %22s\n", "offset", "slow pair, out + d", "slow pair, in + d", "fast pair, out + d", "fast pair, in + d");
    for (size_t d : offs)
        printf("%#12zx %22.4f %22.4f %22.4f %22.4f\n", d, timeit(in[si], out[sj] + d), timeit(in[si] + d, out[sj]), timeit(in[fi], out[fj] + d), timeit(in[fi] + d, out[fj]));
    return 0;
}

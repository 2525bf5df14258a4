// tools/stream_pairs.hip -- the PFAC traffic shape with no PFAC logic: every wave reads 1 KiB of `in` and writes 4 KiB of
// zeros to `out` (non-temporal), small blocks in dispatch order.  Four input and four result allocations, all 16 pairs
// timed: do the placement classes of DESIGN.md 3.3 show up for a bare 1R:4W stream?  Measurement tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void r1w4(const u32x4 *in, i32x4 *out, unsigned *sink)
{
    const size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // 1 KiB of input per wave
    const int lane = threadIdx.x & 63;
    const u32x4 v = in[tile * 64 + lane];
    const i32x4 z = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) __builtin_nontemporal_store(z, &out[tile * 256 + k * 64 + lane]);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = v.x;
}
int main()
{
    const size_t N = size_t(1) << 30;
    unsigned *sink; (void)hipMalloc(&sink, 4);
    void *in[4], *out[4];
    for (int k = 0; k < 4; k++) { (void)hipMalloc(&in[k], N); (void)hipMalloc(&out[k], 4 * N); (void)hipMemset(in[k], k + 1, N); }
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    printf("in : "); for (int k = 0; k < 4; k++) printf("%p ", in[k]); printf("\nout: "); for (int k = 0; k < 4; k++) printf("%p ", out[k]); printf("\n");
    for (int i = 0; i < 4; i++) {
        printf("in %d:", i);
        for (int j = 0; j < 4; j++) {
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(r1w4, dim3(N / 4096), dim3(256), 0, 0, (const u32x4 *)in[i], (i32x4 *)out[j], sink);
            (void)hipEventRecord(a);
            for (int r = 0; r < 10; r++) hipLaunchKernelGGL(r1w4, dim3(N / 4096), dim3(256), 0, 0, (const u32x4 *)in[i], (i32x4 *)out[j], sink);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf(" %.4f", ms / 10);
        }
        printf(" ms\n");
    }
    return 0;
}

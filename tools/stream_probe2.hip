// stream_probe2.hip -- how does the ORDER in which workgroups walk memory change sustained HBM
// bandwidth for the PFAC traffic shape (1 B read : 4 B written)?  Each block owns a contiguous
// chunk of `tilesPerBlock` 1-KiB input tiles (4 KiB of output each); its waves stride through it.
// tilesPerBlock = wavesPerBlock  -> classic non-persistent launch (one tile per wave)
// tilesPerBlock = total/grid     -> persistent blocks with contiguous ranges
// Measurement tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <bool READ, bool NT>
__global__ void stream(const unsigned *in, i32x4 *out, size_t tilesPerBlock, size_t totalTiles, unsigned *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const size_t first = (size_t)blockIdx.x * tilesPerBlock;
    size_t last = first + tilesPerBlock; if (last > totalTiles) last = totalTiles;
    unsigned acc = 0; const i32x4 z = {0, 0, 0, 0};
    for (size_t t = first + wave; t < last; t += wpb) {
        if (READ) { for (int k = 0; k < 4; k++) acc ^= in[t * 256 + k * 64 + lane]; }
        for (int k = 0; k < 4; k++) { if (NT) __builtin_nontemporal_store(z, &out[t * 256 + k * 64 + lane]); else out[t * 256 + k * 64 + lane] = z; }
    }
    if (acc == 0x12345678u) *sink = acc;
}
// interleaved (grid-stride by wave) persistent variant for reference
template <bool READ>
__global__ void stream_gs(const unsigned *in, i32x4 *out, size_t totalTiles, unsigned *sink) {
    const int lane = threadIdx.x & 63; const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t waves = ((size_t)gridDim.x * blockDim.x) >> 6; unsigned acc = 0; const i32x4 z = {0, 0, 0, 0};
    for (size_t t = wave; t < totalTiles; t += waves) {
        if (READ) { for (int k = 0; k < 4; k++) acc ^= in[t * 256 + k * 64 + lane]; }
        for (int k = 0; k < 4; k++) __builtin_nontemporal_store(z, &out[t * 256 + k * 64 + lane]);
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <class F> float timeit(F f, int reps = 7) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); std::vector<float> ms;
    f(); (void)hipDeviceSynchronize();
    for (int i = 0; i < reps; i++) { (void)hipEventRecord(a, 0); f(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b); float t; (void)hipEventElapsedTime(&t, a, b); ms.push_back(t); }
    std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
}
int main() {
    const size_t N = size_t(1) << 30, tiles = N / 1024; unsigned *in; i32x4 *out; unsigned *sink;
    (void)hipMalloc(&in, N); (void)hipMalloc(&out, 4 * N); (void)hipMalloc(&sink, 4); (void)hipMemset(in, 1, N);
    printf("%-10s %-6s %-10s %-8s %10s %10s\n", "kind", "block", "tiles/blk", "grid", "W-only GB/s", "1R4W GB/s(total)");
    for (int block : {256, 512, 1024}) {
        const int wpb = block / 64;
        for (size_t tpb : {(size_t)wpb, (size_t)4 * wpb, (size_t)16 * wpb, (size_t)64 * wpb, (size_t)256 * wpb, (size_t)1024 * wpb}) {
            const size_t grid = (tiles + tpb - 1) / tpb;
            float tw = timeit([&] { hipLaunchKernelGGL((stream<false, true>), dim3(grid), dim3(block), 0, 0, in, out, tpb, tiles, sink); });
            float tr = timeit([&] { hipLaunchKernelGGL((stream<true, true>), dim3(grid), dim3(block), 0, 0, in, out, tpb, tiles, sink); });
            float tp = timeit([&] { hipLaunchKernelGGL((stream<true, false>), dim3(grid), dim3(block), 0, 0, in, out, tpb, tiles, sink); });
            printf("%-10s %-6d %-10zu %-8zu %10.0f %10.0f   plain-store 1R4W %6.0f\n", "chunked", block, tpb, grid, 4.0 * N / tw / 1e6, 5.0 * N / tr / 1e6, 5.0 * N / tp / 1e6);
        }
        for (int mult : {1, 2, 4, 8}) {
            const int grid = 256 * mult * (1024 / block);
            float tw = timeit([&] { hipLaunchKernelGGL((stream_gs<false>), dim3(grid), dim3(block), 0, 0, in, out, tiles, sink); });
            float tr = timeit([&] { hipLaunchKernelGGL((stream_gs<true>), dim3(grid), dim3(block), 0, 0, in, out, tiles, sink); });
            printf("%-10s %-6d %-10s %-8d %10.0f %10.0f\n", "gridstride", block, "-", grid, 4.0 * N / tw / 1e6, 5.0 * N / tr / 1e6);
        }
    }
    float t = timeit([&] { (void)hipMemsetAsync(out, 0, 4 * N, 0); });
    printf("hipMemsetAsync 4 GiB: %.3f ms %.0f GB/s\n", t, 4.0 * N / t / 1e6);
    return 0;
}

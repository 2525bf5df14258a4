// stream_probe.hip -- measures what MI355X HBM actually sustains for the traffic shape of the
// PFAC match path (1 byte read : 4 bytes written), so bench.py's roofline fraction can be read
// against an achievable ceiling as well as the 8 TB/s spec peak.  Measurement tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __global__ __launch_bounds__(1024) void fill_zero(i32x4 *out, size_t n16) {
    const i32x4 z = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(z, &out[i]); else out[i] = z;
    }
}
// 1R:4W -- each wave reads 4 coalesced dwords/lane (1 KiB per wave) and writes 4 KiB, like the scan kernel
template <bool NT> __global__ __launch_bounds__(1024) void read1_write4(const unsigned *in, i32x4 *out, size_t tiles, unsigned *sink) {
    const int lane = threadIdx.x & 63; const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t waves = ((size_t)gridDim.x * blockDim.x) >> 6; unsigned acc = 0;
    for (size_t t = wave; t < tiles; t += waves) {
        unsigned d[4];
        for (int k = 0; k < 4; k++) d[k] = in[t * 256 + k * 64 + lane];
        const i32x4 z = {0, 0, 0, 0};
        for (int k = 0; k < 4; k++) { if (NT) __builtin_nontemporal_store(z, &out[t * 256 + k * 64 + lane]); else out[t * 256 + k * 64 + lane] = z; }
        acc += d[0] ^ d[1] ^ d[2] ^ d[3];
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(1024) void read_only(const u32x4 *in, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { u32x4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(1024) void copy16(const u32x4 *in, u32x4 *out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
template <class F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); std::vector<float> ms;
    f(); f(); hipDeviceSynchronize();
    for (int i = 0; i < reps; i++) { hipEventRecord(a, 0); f(); hipEventRecord(b, 0); hipEventSynchronize(b); float t; hipEventElapsedTime(&t, a, b); ms.push_back(t); }
    std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
}
int main() {
    const size_t N = size_t(1) << 30; unsigned *in; i32x4 *out; unsigned *sink;
    CK(hipMalloc(&in, N)); CK(hipMalloc(&out, 4 * N)); CK(hipMalloc(&sink, 4)); CK(hipMemset(in, 1, N));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz memclock %d kHz bus %d bits\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.memoryBusWidth);
    for (int bpc : {1, 2}) {
        const int grid = p.multiProcessorCount * bpc;
        float t;
        t = timeit([&] { hipLaunchKernelGGL(fill_zero<true>, dim3(grid), dim3(1024), 0, 0, out, 4 * N / 16); });
        printf("grid %4d fill_zero nt      4 GiB: %.3f ms  %.0f GB/s written\n", grid, t, 4.0 * N / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(fill_zero<false>, dim3(grid), dim3(1024), 0, 0, out, 4 * N / 16); });
        printf("grid %4d fill_zero plain   4 GiB: %.3f ms  %.0f GB/s written\n", grid, t, 4.0 * N / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(read1_write4<true>, dim3(grid), dim3(1024), 0, 0, in, out, N / 1024, sink); });
        printf("grid %4d read1_write4 nt   5 GiB: %.3f ms  %.0f GB/s total, %.0f GB/s input\n", grid, t, 5.0 * N / t / 1e6, 1.0 * N / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(read1_write4<false>, dim3(grid), dim3(1024), 0, 0, in, out, N / 1024, sink); });
        printf("grid %4d read1_write4 plain5 GiB: %.3f ms  %.0f GB/s total, %.0f GB/s input\n", grid, t, 5.0 * N / t / 1e6, 1.0 * N / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(read_only, dim3(grid), dim3(1024), 0, 0, (const u32x4 *)out, 4 * N / 16, sink); });
        printf("grid %4d read_only         4 GiB: %.3f ms  %.0f GB/s read\n", grid, t, 4.0 * N / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(1024), 0, 0, (const u32x4 *)out, (u32x4 *)out + (2 * N / 16), 2 * N / 16); });
        printf("grid %4d copy 2 GiB->2 GiB      : %.3f ms  %.0f GB/s total\n", grid, t, 4.0 * N / t / 1e6);
    }
    float t = timeit([&] { hipMemsetAsync(out, 0, 4 * N, 0); });
    printf("hipMemsetAsync 4 GiB: %.3f ms %.0f GB/s\n", t, 4.0 * N / t / 1e6);
    return 0;
}

// stream_probe3.hip -- why does a plain framework fill of 4 GiB finish in ~0.3 ms (14 TB/s apparent)?
// Replicates that launch shape and varies value / block work to find what the memory system rewards.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int PER_THREAD>   // 16-byte stores per thread; block = 256 threads; consecutive threads -> consecutive 16 B
__global__ __launch_bounds__(256) void fill_small(i32x4 *out, int value, int mixLane) {
    const size_t base = (size_t)blockIdx.x * 256 * PER_THREAD + threadIdx.x;
    const int v = mixLane ? value ^ (int)(threadIdx.x * 2654435761u) ^ (int)blockIdx.x : value;
    const i32x4 x = {v, v, v, v};
#pragma unroll
    for (int j = 0; j < PER_THREAD; j++) out[base + (size_t)j * 256] = x;
}
template <class F> float timeit(F f, int reps = 7) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); std::vector<float> ms;
    f(); (void)hipDeviceSynchronize();
    for (int i = 0; i < reps; i++) { (void)hipEventRecord(a, 0); f(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b); float t; (void)hipEventElapsedTime(&t, a, b); ms.push_back(t); }
    std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
}
int main() {
    const size_t BYTES = size_t(4) << 30; i32x4 *out; (void)hipMalloc(&out, BYTES);
    auto run = [&](const char *name, auto kernel, int per, int value, int mix) {
        const size_t blocks = BYTES / 16 / 256 / per;
        float t = timeit([&] { hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, value, mix); });
        printf("%-28s value=%11d mix=%d blocks=%8zu  %.3f ms  %.0f GB/s\n", name, value, mix, blocks, t, BYTES / t / 1e6);
    };
    for (int mix : {0, 1}) for (int value : {0, -1, 0x12345678}) {
        run("fill_small<1>", fill_small<1>, 1, value, mix);
        run("fill_small<2>", fill_small<2>, 2, value, mix);
        run("fill_small<4>", fill_small<4>, 4, value, mix);
        run("fill_small<16>", fill_small<16>, 16, value, mix);
    }
    return 0;
}

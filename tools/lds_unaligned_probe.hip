// tools/lds_unaligned_probe.hip -- does ds_read_b32 at a byte-unaligned LDS address return the 4 bytes
// at that address on gfx950?  (dev tool, GPU box only)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef const __attribute__((address_space(3))) uint32_t LdsWordU __attribute__((aligned(1)));
__global__ void k(uint32_t *out)
{
    extern __shared__ unsigned char smem[];
    for (int i = threadIdx.x; i < 1024; i += 64) smem[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const uint32_t a = threadIdx.x * 5 + 1;                   // mostly unaligned
    out[threadIdx.x] = *reinterpret_cast<LdsWordU *>((uintptr_t)a);
}
int main()
{
    uint32_t *d, h[64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; t++) {
        const uint32_t a = t * 5 + 1;
        uint32_t want = 0;
        for (int b = 0; b < 4; b++) want |= (uint32_t)(unsigned char)((a + b) * 7 + 3) << (8 * b);
        if (h[t] != want) { if (bad < 4) printf("lane %d addr %u got %08x want %08x\n", t, a, h[t], want); bad++; }
    }
    printf("unaligned ds_read_b32: %s (%d of 64 lanes differ)\n", bad ? "NOT byte-exact" : "OK", bad);
    return 0;
}

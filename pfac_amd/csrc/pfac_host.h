/*
 * pfac_host.h -- internal declarations of the host side of libpfac.so.
 */
#ifndef PFAC_HOST_H_
#define PFAC_HOST_H_

#include "pfac_context.h"

namespace pfac {

/* pattern_compiler.cpp */
PFAC_status_t compilePatternFile(const char *filename, Automaton &fa, unsigned int flags = 0);
PFAC_status_t compilePatternBytes(std::vector<unsigned char> bytes, Automaton &fa, unsigned int flags = 0);   /* flags: PFACX_READ_* */
void buildInitialRow(const Automaton &fa, std::vector<int> &row);
void buildFilter(const Automaton &fa, Filter &f);
void buildReduceFilter(const Automaton &fa, Filter &f);     /* gram1 + prefix4 alone (buildFilter calls it; a compiled set is loaded without them) */

/* tables.cpp */
PFAC_status_t buildDenseTable(const Automaton &fa, std::vector<int> &dense);
PFAC_status_t buildHashTable(const Automaton &fa, std::vector<Int2> &rowPtr,
                             std::vector<Int2> &valPtr);

PFAC_status_t buildChainedHashTable(const Automaton &fa, std::vector<ChainSlot> &slots, int &jumpLog2, bool narrow = false);

/* cpu_engine.cpp: PFAC_PLATFORM_CPU / PFAC_PLATFORM_CPU_OMP */
PFAC_status_t matchOnCpu(const PFAC_context *ctx, const unsigned char *in, size_t n, int *out,
                         bool useOpenMP);

} // namespace pfac

/* ---- internals of libpfac.so shared by pfac_api.cpp, host_pipeline.cpp, multi_gpu.cpp, compiled_set.cpp ---- */
#include <hip/hip_runtime_api.h>
namespace pfac_internal {

template <class T>
void devFree(T *&p)
{
    if (p) { (void)hipFree(p); p = nullptr; }
}

template <class T>
PFAC_status_t upload(T *&dst, const T *src, size_t count)
{
    const size_t bytes = (count ? count : 1) * sizeof(T);
    if (hipMalloc(reinterpret_cast<void **>(&dst), bytes) != hipSuccess) {
        dst = nullptr;
        (void)hipGetLastError();
        return PFAC_STATUS_CUDA_ALLOC_FAILED;
    }
    if (count && hipMemcpy(dst, src, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        devFree(dst);
        return PFAC_STATUS_INTERNAL_ERROR;
    }
    return PFAC_STATUS_SUCCESS;
}

/* positions per piece of the pipelined PFAC_matchFromHost: 32 Mi positions = 32 MiB up, 128 MiB down */
constexpr size_t kHostPiece = size_t(32) << 20;

/* pfac_api.cpp */
void freeTables(PFAC_context *c);
void freeHostStage(PFAC_context *c);
void freeResources(PFAC_context *c);
PFAC_status_t bindTable(PFAC_context *c);
PFAC_status_t bindCommon(PFAC_context *c, bool build = true);
void correctTextureMode(PFAC_context *c);
PFAC_status_t matchHostOnCpuPlatform(PFAC_context *c, const char *in, size_t n, int *out);
PFAC_status_t prepareCpuPlatformLocked(PFAC_context *c);                                        /* the caller holds c->lock */
PFAC_status_t matchHostOnCpuPlatformPrepared(PFAC_context *c, const char *in, size_t n, int *out);   /* ... has called the above; any number of threads */
/* host_pipeline.cpp: the caller holds c->lock */
PFAC_status_t prepareHostPath(PFAC_context *c, size_t maxBytes);
PFAC_status_t matchDeviceLocked(PFAC_context *c, char *d_inputString, size_t size, int *d_matched_result);
PFAC_status_t matchHostOnGpu(PFAC_context *c, char *h_inputString, size_t owned, size_t readable, int *h_matched_result);
PFAC_status_t matchHostReduceOnGpu(PFAC_context *c, char *h_inputString, size_t size, size_t readable, size_t posBase, int *h_matched_result, int *h_pos, int *h_num_matched);

} // namespace pfac_internal

#endif

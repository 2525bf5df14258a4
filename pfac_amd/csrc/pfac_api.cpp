/*
 * pfac_api.cpp -- the reference-compatible C ABI of libpfac.so (include/PFAC.h)
 * plus the PFACX_ extensions (include/pfac_ext.h).
 *
 * Mirrors the observable behaviour of PFAC/src/PFAC.cpp: argument-check order,
 * status codes, lifecycle (a second readPatternFromFile replaces the first,
 * setPerfMode after load rebuilds the table), the dlopen'd kernel-module seam
 * and the per-call device temporaries of PFAC_matchFromHost.  HIP replaces the
 * CUDA runtime 1:1 on the host side; all device work is in the module
 * (scan_gfx950.hip).
 */
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "pfac_host.h"

using pfac::Int2;

/* positions per piece of the pipelined PFAC_matchFromHost: 32 Mi positions = 32 MiB up, 128 MiB down */
static constexpr size_t kHostPiece = size_t(32) << 20;

namespace {

template <class T>
void devFree(T *&p)
{
    if (p) { (void)hipFree(p); p = nullptr; }
}

/* ref PFAC_freeTable, PFAC.cpp:256-297 (perfMode-dependent tables only) */
void freeTables(PFAC_context *c)
{
    std::vector<int>().swap(c->h_dense);
    std::vector<Int2>().swap(c->h_hashRow);
    std::vector<Int2>().swap(c->h_hashVal);
    devFree(c->d_dense);
    devFree(c->d_hashRow);
    devFree(c->d_hashVal);
    devFree(c->d_chainSlots);
    devFree(c->d_rootSlots);
    c->numChainSlots = 0;
    c->numOfTableEntry = c->sizeOfTableEntry = c->sizeOfTableInBytes = 0;
}

void freeHostStage(PFAC_context *c)
{
    for (int b = 0; b < 2; b++) {
        devFree(c->d_stageIn[b]);
        devFree(c->d_stageOut[b]);
        if (c->evUp[b]) (void)hipEventDestroy(static_cast<hipEvent_t>(c->evUp[b]));
        if (c->evScan[b]) (void)hipEventDestroy(static_cast<hipEvent_t>(c->evScan[b]));
        if (c->evDown[b]) (void)hipEventDestroy(static_cast<hipEvent_t>(c->evDown[b]));
        c->evUp[b] = c->evScan[b] = c->evDown[b] = nullptr;
    }
    if (c->stageUp) (void)hipStreamDestroy(static_cast<hipStream_t>(c->stageUp));
    if (c->stageDown) (void)hipStreamDestroy(static_cast<hipStream_t>(c->stageDown));
    c->stageUp = c->stageDown = nullptr;
    c->hostStagePositions = 0;
}

/* ref PFAC_freeResource, PFAC.cpp:221-254 */
void freeResources(PFAC_context *c)
{
    freeTables(c);
    std::vector<int>().swap(c->h_initialRow);
    devFree(c->d_initialRow);
    devFree(c->d_gram3);
    devFree(c->d_shortBits);
    devFree(c->d_gram4);
    devFree(c->d_reduceCount);
    devFree(c->d_workCounters);
    devFree(c->d_reduceScratch);
    c->reduceScratchBytes = 0;
    freeHostStage(c);
    devFree(c->d_final3);
    c->fa = pfac::Automaton();
    c->filter = pfac::Filter();
    c->isPatternsReady = false;
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

/* Upload the chained device form of the hashed table (tables.cpp: buildChainedHashTable): what the scan kernel
 * walks in BOTH perf modes.  In PFAC_TIME_DRIVEN mode the hashed layout it derives from is built here and
 * dropped again: the handle's reference-layout table (PFACX_getTable, the dump, the simple kernel) stays dense. */
PFAC_status_t uploadChainedHashTable(PFAC_context *c)
{
    std::vector<pfac::ChainSlot> slots, root;
    PFAC_status_t st;
    if (c->perfMode == PFAC_SPACE_DRIVEN) {
        st = pfac::buildChainedHashTable(c->fa, c->h_hashRow, c->h_hashVal, slots, root);
    } else {
        std::vector<Int2> rowPtr, valPtr;
        st = pfac::buildHashTable(c->fa, rowPtr, valPtr);
        if (st == PFAC_STATUS_SUCCESS) st = pfac::buildChainedHashTable(c->fa, rowPtr, valPtr, slots, root);
    }
    if (st != PFAC_STATUS_SUCCESS) return st;
    c->numChainSlots = slots.size();
    st = upload(c->d_chainSlots, slots.data(), slots.size());
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_rootSlots, root.data(), root.size());
    return st;
}

/* ref PFAC_bindTable -> PFAC_create2DTable / PFAC_createHashTable, PFAC.cpp:321-648 */
PFAC_status_t bindTable(PFAC_context *c)
{
    if (!c->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    PFAC_status_t st;
    if (c->perfMode == PFAC_TIME_DRIVEN) {
        if (c->h_dense.empty()) {
            st = pfac::buildDenseTable(c->fa, c->h_dense);
            if (st != PFAC_STATUS_SUCCESS) return st;
        }
        c->numOfTableEntry = (size_t)pfac::kCharSet * (size_t)c->fa.numStates;
        c->sizeOfTableEntry = sizeof(int);
        c->sizeOfTableInBytes = c->numOfTableEntry * c->sizeOfTableEntry;
        if (c->hasDevice && !c->d_dense) {
            st = upload(c->d_dense, c->h_dense.data(), c->h_dense.size());
            if (st == PFAC_STATUS_SUCCESS) st = uploadChainedHashTable(c);
            if (st != PFAC_STATUS_SUCCESS) { freeTables(c); return st; }
        }
    } else {
        if (c->h_hashRow.empty()) {
            st = pfac::buildHashTable(c->fa, c->h_hashRow, c->h_hashVal);
            if (st != PFAC_STATUS_SUCCESS) return st;
        }
        c->numOfTableEntry = c->h_hashVal.size();
        c->sizeOfTableEntry = sizeof(Int2);
        c->sizeOfTableInBytes = c->numOfTableEntry * c->sizeOfTableEntry;
        if (c->hasDevice && !c->d_hashRow) {
            st = upload(c->d_hashRow, c->h_hashRow.data(), c->h_hashRow.size());
            if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_hashVal, c->h_hashVal.data(), c->h_hashVal.size());
            if (st == PFAC_STATUS_SUCCESS) st = uploadChainedHashTable(c);
            if (st != PFAC_STATUS_SUCCESS) { freeTables(c); return st; }
        }
    }
    return PFAC_STATUS_SUCCESS;
}

/* tables that do not depend on perfMode: initial-state row and prefilter */
PFAC_status_t bindCommon(PFAC_context *c)
{
    pfac::buildInitialRow(c->fa, c->h_initialRow);
    pfac::buildFilter(c->fa, c->filter);
    if (!c->hasDevice) return PFAC_STATUS_SUCCESS;
    PFAC_status_t st = upload(c->d_initialRow, c->h_initialRow.data(), c->h_initialRow.size());
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_gram3, c->filter.gram3.data(), c->filter.gram3.size());
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_shortBits, c->filter.shortBits.data(), c->filter.shortBits.size());
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_gram4, c->filter.gram4.data(), c->filter.gram4.size());
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_final3, c->filter.final3.data(), c->filter.final3.size());
    const unsigned int zero = 0;
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_reduceCount, &zero, 1);
    if (st == PFAC_STATUS_SUCCESS) {               /* chunk counters of the scan kernel, reset before every launch */
        const std::vector<unsigned int> zeros(pfac::kWorkCounterWords, 0u);
        st = upload(c->d_workCounters, zeros.data(), zeros.size());
    }
    return st;
}

/* ref correctTextureMode, PFAC.cpp:819-833: AUTOMATIC is resolved (and stored) at match time */
void correctTextureMode(PFAC_context *c)
{
    if (c->textureMode == PFAC_AUTOMATIC)
        c->textureMode = (c->numOfTableEntry < pfac::kTexMaxEntries) ? PFAC_TEXTURE_ON : PFAC_TEXTURE_OFF;
}

/* directory that holds this shared object, so the module is found next to it
 * without LD_LIBRARY_PATH (which still works, as in the reference README:96-103) */
std::string selfDirectory()
{
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&selfDirectory), &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t slash = p.rfind('/');
        if (slash != std::string::npos) return p.substr(0, slash + 1);
    }
    return std::string();
}

PFAC_status_t loadModule(PFAC_context *c)
{
    const std::string name = "libpfac_" + c->archName + ".so";
    void *m = dlopen((selfDirectory() + name).c_str(), RTLD_NOW);
    if (!m) m = dlopen(name.c_str(), RTLD_NOW);
    if (!m) return PFAC_STATUS_LIB_NOT_EXIST;
    c->module = m;
    c->kernel_time_driven_ptr = (PFAC_kernel_protoType)dlsym(m, "PFAC_kernel_timeDriven_warpper");
    c->kernel_space_driven_ptr = (PFAC_kernel_protoType)dlsym(m, "PFAC_kernel_spaceDriven_warpper");
    c->reduce_kernel_ptr = (PFAC_reduce_kernel_protoType)dlsym(m, "PFAC_reduce_kernel");
    c->reduce_inplace_kernel_ptr = (PFAC_reduce_kernel_protoType)dlsym(m, "PFAC_reduce_inplace_kernel");
    if (!c->kernel_time_driven_ptr || !c->kernel_space_driven_ptr || !c->reduce_kernel_ptr ||
        !c->reduce_inplace_kernel_ptr)
        return PFAC_STATUS_INTERNAL_ERROR;
    return PFAC_STATUS_SUCCESS;
}

/* ref the CPU branch of matchFromHost / matchFromHostReduce, PFAC.cpp:899-913 */
PFAC_status_t matchHostOnCpuPlatform(PFAC_context *c, const char *in, size_t n, int *out)
{
    bool omp = false;
    if (c->platform == PFAC_PLATFORM_CPU_OMP) omp = (std::getenv("OMP_NUM_THREADS") != nullptr);
    return pfac::matchOnCpu(c, reinterpret_cast<const unsigned char *>(in), n, out, omp);
}

} // namespace

extern "C" {

PFAC_status_t PFAC_create(PFAC_handle_t *handle)
{
    if (!handle) return PFAC_STATUS_INVALID_PARAMETER;
    PFAC_context *c = new (std::nothrow) PFAC_context();
    *handle = c;
    if (!c) return PFAC_STATUS_ALLOC_FAILED;

    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (PFAC_status_t)e;          /* ref PFAC.cpp:148-151 */
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (PFAC_status_t)e;
    c->device = dev;
    c->multiProcessorCount = prop.multiProcessorCount;
    c->archName = prop.gcnArchName;                        /* "gfx950:sramecc+:xnack-" */
    const size_t colon = c->archName.find(':');
    if (colon != std::string::npos) c->archName.resize(colon);
    c->hasDevice = true;
    return loadModule(c);
}

PFAC_status_t PFACX_createHostOnly(PFAC_handle_t *handle)
{
    if (!handle) return PFAC_STATUS_INVALID_PARAMETER;
    PFAC_context *c = new (std::nothrow) PFAC_context();
    *handle = c;
    if (!c) return PFAC_STATUS_ALLOC_FAILED;
    c->hasDevice = false;
    c->platform = PFAC_PLATFORM_CPU;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFAC_destroy(PFAC_handle_t handle)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    freeResources(handle);
    /* the module stays mapped: other handles may share it (dlopen refcounts) */
    if (handle->module) dlclose(handle->module);
    delete handle;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFAC_setPlatform(PFAC_handle_t handle, PFAC_platform_t platform)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (platform != PFAC_PLATFORM_GPU && platform != PFAC_PLATFORM_CPU && platform != PFAC_PLATFORM_CPU_OMP)
        return PFAC_STATUS_INVALID_PARAMETER;
    handle->platform = (int)platform;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFAC_setTextureMode(PFAC_handle_t handle, PFAC_textureMode_t textureModeSel)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (textureModeSel != PFAC_AUTOMATIC && textureModeSel != PFAC_TEXTURE_ON && textureModeSel != PFAC_TEXTURE_OFF)
        return PFAC_STATUS_INVALID_PARAMETER;
    handle->textureMode = (int)textureModeSel;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFAC_setPerfMode(PFAC_handle_t handle, PFAC_perfMode_t perfModeSel)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (perfModeSel != PFAC_TIME_DRIVEN && perfModeSel != PFAC_SPACE_DRIVEN) return PFAC_STATUS_INVALID_PARAMETER;
    const bool rebuild = handle->isPatternsReady && (int)perfModeSel != handle->perfMode;
    handle->perfMode = (int)perfModeSel;
    if (rebuild) {                                         /* ref PFAC.cpp:794-814 */
        freeTables(handle);
        PFAC_status_t st = bindTable(handle);
        if (st != PFAC_STATUS_SUCCESS) { freeTables(handle); return st; }
    }
    return PFAC_STATUS_SUCCESS;
}

const char *PFAC_getErrorString(PFAC_status_t status)
{
    if (status == PFAC_STATUS_SUCCESS) return "PFAC_STATUS_SUCCESS: operation is successful";
    if ((int)status < (int)PFAC_STATUS_BASE) return hipGetErrorString((hipError_t)status);
    switch (status) {
    case PFAC_STATUS_ALLOC_FAILED: return "PFAC_STATUS_ALLOC_FAILED: allocation fails on host memory";
    case PFAC_STATUS_CUDA_ALLOC_FAILED: return "PFAC_STATUS_CUDA_ALLOC_FAILED: allocation fails on device memory";
    case PFAC_STATUS_INVALID_HANDLE: return "PFAC_STATUS_INVALID_HANDLE: handle is invalid (NULL)";
    case PFAC_STATUS_INVALID_PARAMETER: return "PFAC_STATUS_INVALID_PARAMETER: parameter is invalid";
    case PFAC_STATUS_PATTERNS_NOT_READY: return "PFAC_STATUS_PATTERNS_NOT_READY: please call PFAC_readPatternFromFile() first";
    case PFAC_STATUS_FILE_OPEN_ERROR: return "PFAC_STATUS_FILE_OPEN_ERROR: pattern file does not exist";
    case PFAC_STATUS_LIB_NOT_EXIST: return "PFAC_STATUS_LIB_NOT_EXIST: cannot find PFAC library, please check LD_LIBRARY_PATH";
    case PFAC_STATUS_ARCH_MISMATCH: return "PFAC_STATUS_ARCH_MISMATCH: sm1.0 is not supported";
    case PFAC_STATUS_MUTEX_ERROR: return "PFAC_STATUS_MUTEX_ERROR: please report bugs. Workaround: choose non-texture mode.";
    default: return "PFAC_STATUS_INTERNAL_ERROR: please report bugs";
    }
}

/* Text format of the reference (PFAC.cpp:1188-1246, user guide r1.2 p.21). */
PFAC_status_t PFAC_dumpTransitionTable(PFAC_handle_t handle, FILE *fp)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    if (!fp) fp = stdout;
    const pfac::Automaton &fa = handle->fa;
    std::fprintf(fp, "# Transition table: number of states = %d, initial state = %d\n", fa.numStates, fa.initialState);
    std::fprintf(fp, "# (current state, input character) -> next state \n");
    for (int s = 0; s < fa.numStates; s++) {
        for (int e = fa.edgeBegin[s]; e < fa.edgeBegin[s + 1]; e++) {
            const int ch = fa.edgeCh[e];
            if (ch >= 32 && ch <= 126) std::fprintf(fp, "(%4d,%4c) -> %d \n", s, ch, fa.edgeNext[e]);
            else std::fprintf(fp, "(%4d,%4.2x) -> %d \n", s, ch, fa.edgeNext[e]);
        }
    }
    std::fprintf(fp, "# Output table: number of final states = %d\n", fa.numPatterns);
    std::fprintf(fp, "# [final state] [matched pattern ID] [pattern length] [pattern(string literal)] \n");
    for (int id = 1; id <= fa.numPatterns; id++) {
        std::fprintf(fp, "%5d %5d %5d    \"", id, id, fa.patternLen[id]);
        const unsigned char *p = fa.file.data() + fa.patternOff[id];
        for (int i = 0; i < fa.patternLen[id]; i++) {
            if (p[i] >= 32 && p[i] <= 126) std::fputc(p[i], fp);
            else std::fprintf(fp, "%2.2x", (int)p[i]);
        }
        std::fprintf(fp, "\"\n");
    }
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFAC_readPatternFromFile(PFAC_handle_t handle, char *filename)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!filename) return PFAC_STATUS_INVALID_PARAMETER;
    if (handle->isPatternsReady) freeResources(handle);            /* ref PFAC.cpp:663-666 */
    if (std::strlen(filename) >= (size_t)pfac::kFileNameLen) return PFAC_STATUS_INTERNAL_ERROR;  /* ref :668-672 */
    handle->patternFile = filename;

    PFAC_status_t st = pfac::compilePatternFile(filename, handle->fa);
    if (st != PFAC_STATUS_SUCCESS) { freeResources(handle); return st; }
    handle->isPatternsReady = true;
    st = bindCommon(handle);
    if (st == PFAC_STATUS_SUCCESS) st = bindTable(handle);
    if (st != PFAC_STATUS_SUCCESS) { freeResources(handle); return st; }
    return PFAC_STATUS_SUCCESS;
}

/* pfac_ext.h: the same pattern-file bytes from memory instead of from a file */
PFAC_status_t PFACX_readPatternFromMemory(PFAC_handle_t handle, const char *patterns, size_t size)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!patterns && size) return PFAC_STATUS_INVALID_PARAMETER;
    if (handle->isPatternsReady) freeResources(handle);
    handle->patternFile.clear();
    PFAC_status_t st;
    try {
        st = pfac::compilePatternBytes(std::vector<unsigned char>(patterns, patterns + size), handle->fa);
    } catch (const std::bad_alloc &) { st = PFAC_STATUS_ALLOC_FAILED; }
    if (st != PFAC_STATUS_SUCCESS) { freeResources(handle); return st; }
    handle->isPatternsReady = true;
    st = bindCommon(handle);
    if (st == PFAC_STATUS_SUCCESS) st = bindTable(handle);
    if (st != PFAC_STATUS_SUCCESS) { freeResources(handle); return st; }
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFAC_matchFromDevice(PFAC_handle_t handle, char *d_inputString, size_t size, int *d_matched_result)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;                /* check order: ref PFAC.cpp:846-861 */
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    if (!d_inputString) return PFAC_STATUS_INVALID_PARAMETER;
    if (!d_matched_result) return PFAC_STATUS_INVALID_PARAMETER;
    if (size == 0) return PFAC_STATUS_SUCCESS;
    if (!handle->hasDevice || !handle->module) return PFAC_STATUS_LIB_NOT_EXIST;   /* never a CPU fallback */
    correctTextureMode(handle);
    if (handle->perfMode == PFAC_TIME_DRIVEN)
        return handle->kernel_time_driven_ptr(handle, d_inputString, size, d_matched_result);
    if (handle->perfMode == PFAC_SPACE_DRIVEN)
        return handle->kernel_space_driven_ptr(handle, d_inputString, size, d_matched_result);
    return PFAC_STATUS_INTERNAL_ERROR;
}

PFAC_status_t PFAC_matchFromHost(PFAC_handle_t handle, char *h_inputString, size_t size, int *h_matched_result)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    if (!h_inputString) return PFAC_STATUS_INVALID_PARAMETER;
    if (!h_matched_result) return PFAC_STATUS_INVALID_PARAMETER;
    if (size == 0) return PFAC_STATUS_SUCCESS;
    if (handle->platform != PFAC_PLATFORM_GPU)
        return matchHostOnCpuPlatform(handle, h_inputString, size, h_matched_result);
    if (!handle->hasDevice || !handle->module) return PFAC_STATUS_LIB_NOT_EXIST;

    /*
     * The reference allocates, uploads, scans, downloads and frees in sequence (PFAC.cpp:916-960), which
     * leaves the scan idle for the 5 bytes per position that cross the host link.  Here the stream is
     * cut into pieces of kHostPiece positions: piece i+1 is uploaded and piece i-1 downloaded while
     * piece i is scanned.  Each piece is scanned together with the maxPatternLen
     * bytes behind it -- a walk may read that far -- and only its own results go back
     * (omp_PFAC.cpp:324,377).  The staging buffers, two copy streams and their events belong to the
     * handle and are created on first use; the scan itself stays on the default stream.
     */
    PFAC_context *c = handle;
    const size_t overlap = (size_t)c->fa.maxPatternLen;
    const size_t piece = size < kHostPiece ? size : kHostPiece;
    const size_t need = piece + overlap;
    if (c->hostStagePositions < need) {
        freeHostStage(c);
        bool ok = true;
        for (int b = 0; b < 2 && ok; b++) {
            ok = hipMalloc(reinterpret_cast<void **>(&c->d_stageIn[b]), (need + 3) & ~size_t(3)) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void **>(&c->d_stageOut[b]), need * sizeof(int)) == hipSuccess;
            hipEvent_t e[3] = {nullptr, nullptr, nullptr};
            for (int k = 0; k < 3 && ok; k++) ok = hipEventCreateWithFlags(&e[k], hipEventDisableTiming) == hipSuccess;
            c->evUp[b] = e[0]; c->evScan[b] = e[1]; c->evDown[b] = e[2];
        }
        hipStream_t up = nullptr, down = nullptr;
        ok = ok && hipStreamCreateWithFlags(&up, hipStreamNonBlocking) == hipSuccess &&
             hipStreamCreateWithFlags(&down, hipStreamNonBlocking) == hipSuccess;
        c->stageUp = up; c->stageDown = down;
        if (!ok) { (void)hipGetLastError(); freeHostStage(c); return PFAC_STATUS_CUDA_ALLOC_FAILED; }
        c->hostStagePositions = need;
    }
    hipStream_t up = static_cast<hipStream_t>(c->stageUp), down = static_cast<hipStream_t>(c->stageDown);
    PFAC_status_t st = PFAC_STATUS_SUCCESS;
    bool used[2] = {false, false};
    size_t i = 0;
    for (size_t off = 0; off < size && st == PFAC_STATUS_SUCCESS; off += piece, i++) {
        const int b = (int)(i & 1);
        const size_t owned = size - off < piece ? size - off : piece;
        const size_t scanned = size - off < owned + overlap ? size - off : owned + overlap;
        hipEvent_t evUp = static_cast<hipEvent_t>(c->evUp[b]), evScan = static_cast<hipEvent_t>(c->evScan[b]),
                   evDown = static_cast<hipEvent_t>(c->evDown[b]);
        bool ok = true;
        if (used[b]) ok = hipStreamWaitEvent(up, evScan, 0) == hipSuccess;          /* the scan of piece i-2 has read this buffer */
        ok = ok && hipMemcpyAsync(c->d_stageIn[b], h_inputString + off, scanned, hipMemcpyHostToDevice, up) == hipSuccess &&
             hipEventRecord(evUp, up) == hipSuccess && hipStreamWaitEvent(nullptr, evUp, 0) == hipSuccess;
        if (ok && used[b]) ok = hipStreamWaitEvent(nullptr, evDown, 0) == hipSuccess;   /* its results have left this buffer */
        if (!ok) { st = PFAC_STATUS_INTERNAL_ERROR; break; }
        st = PFAC_matchFromDevice(handle, c->d_stageIn[b], scanned, c->d_stageOut[b]);
        if (st != PFAC_STATUS_SUCCESS) break;
        ok = hipEventRecord(evScan, nullptr) == hipSuccess && hipStreamWaitEvent(down, evScan, 0) == hipSuccess &&
             hipMemcpyAsync(h_matched_result + off, c->d_stageOut[b], owned * sizeof(int), hipMemcpyDeviceToHost, down) == hipSuccess &&
             hipEventRecord(evDown, down) == hipSuccess;
        if (!ok) st = PFAC_STATUS_INTERNAL_ERROR;
        used[b] = true;
    }
    const bool drained = hipStreamSynchronize(up) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess &&
                         hipStreamSynchronize(down) == hipSuccess;
    if (!drained && st == PFAC_STATUS_SUCCESS) st = PFAC_STATUS_INTERNAL_ERROR;
    return st;
}

PFAC_status_t PFAC_matchFromDeviceReduce(PFAC_handle_t handle, char *d_inputString, size_t size,
                                         int *d_matched_result, int *d_pos, int *h_num_matched)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    if (!d_inputString || !d_matched_result || !d_pos || !h_num_matched) return PFAC_STATUS_INVALID_PARAMETER;
    if (size == 0) return PFAC_STATUS_SUCCESS;
    if (!handle->hasDevice || !handle->module) return PFAC_STATUS_LIB_NOT_EXIST;
    if (size > (size_t)0x7fffffff) return PFAC_STATUS_INVALID_PARAMETER;   /* int positions */
    correctTextureMode(handle);
    PFAC_reduce_kernel_protoType fn =
        handle->perfMode == PFAC_TIME_DRIVEN ? handle->reduce_kernel_ptr : handle->reduce_inplace_kernel_ptr;
    return fn(handle, reinterpret_cast<int *>(d_inputString), (int)size, d_matched_result, d_pos, h_num_matched,
              nullptr, nullptr);
}

PFAC_status_t PFAC_matchFromHostReduce(PFAC_handle_t handle, char *h_inputString, size_t size,
                                       int *h_matched_result, int *h_pos, int *h_num_matched)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    if (!h_inputString || !h_matched_result || !h_pos || !h_num_matched) return PFAC_STATUS_INVALID_PARAMETER;
    if (size == 0) return PFAC_STATUS_SUCCESS;
    if (size > (size_t)0x7fffffff) return PFAC_STATUS_INVALID_PARAMETER;

    if (handle->platform != PFAC_PLATFORM_GPU) {                  /* ref PFAC.cpp:1036-1068 */
        PFAC_status_t st = matchHostOnCpuPlatform(handle, h_inputString, size, h_matched_result);
        if (st != PFAC_STATUS_SUCCESS) return st;
        int z = 0;
        for (size_t i = 0; i < size; i++) {
            const int m = h_matched_result[i];
            if (m > 0) { h_matched_result[z] = m; h_pos[z] = (int)i; z++; }
        }
        *h_num_matched = z;
        return PFAC_STATUS_SUCCESS;
    }
    if (!handle->hasDevice || !handle->module) return PFAC_STATUS_LIB_NOT_EXIST;

    char *d_in = nullptr;
    int *d_out = nullptr, *d_pos = nullptr;
    const size_t inBytes = (size + 3) & ~size_t(3);
    hipError_t e1 = hipMalloc(reinterpret_cast<void **>(&d_in), inBytes);
    hipError_t e2 = hipMalloc(reinterpret_cast<void **>(&d_out), size * sizeof(int));
    hipError_t e3 = hipMalloc(reinterpret_cast<void **>(&d_pos), size * sizeof(int));
    PFAC_status_t st = PFAC_STATUS_SUCCESS;
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        (void)hipGetLastError();
        st = PFAC_STATUS_CUDA_ALLOC_FAILED;
    }
    if (st == PFAC_STATUS_SUCCESS && hipMemcpy(d_in, h_inputString, size, hipMemcpyHostToDevice) != hipSuccess)
        st = PFAC_STATUS_INTERNAL_ERROR;
    if (st == PFAC_STATUS_SUCCESS) {
        correctTextureMode(handle);
        PFAC_reduce_kernel_protoType fn =
            handle->perfMode == PFAC_TIME_DRIVEN ? handle->reduce_kernel_ptr : handle->reduce_inplace_kernel_ptr;
        st = fn(handle, reinterpret_cast<int *>(d_in), (int)size, d_out, d_pos, h_num_matched, h_matched_result, h_pos);
    }
    if (e1 == hipSuccess) (void)hipFree(d_in);
    if (e2 == hipSuccess) (void)hipFree(d_out);
    if (e3 == hipSuccess) (void)hipFree(d_pos);
    return st;
}

/* ------------------------------------------------------------- extensions */

PFAC_status_t PFACX_getInfo(PFAC_handle_t handle, PFACX_info_t *info)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!info) return PFAC_STATUS_INVALID_PARAMETER;
    std::memset(info, 0, sizeof(*info));
    info->numOfPatterns = handle->fa.numPatterns;
    info->numOfStates = handle->fa.numStates;
    info->numOfFinalStates = handle->fa.numPatterns;
    info->initialState = handle->fa.initialState;
    info->maxPatternLen = handle->fa.maxPatternLen;
    info->numOfLeaves = handle->fa.numLeaves;
    info->perfMode = handle->perfMode;
    info->textureMode = handle->textureMode;
    info->platform = handle->platform;
    info->hasDevice = handle->hasDevice ? 1 : 0;
    info->numOfTableEntry = handle->numOfTableEntry;
    info->sizeOfTableEntry = handle->sizeOfTableEntry;
    info->sizeOfTableInBytes = handle->sizeOfTableInBytes;
    info->filterLog2Bits = handle->filter.log2Bits;
    info->filterHasShort = handle->filter.hasShort ? 1 : 0;
    info->filterBitsSet = handle->filter.bitsSet;
    info->kernelVariant = handle->kernelVariant;
    info->filterLog2Bits4 = handle->filter.log2Bits4;
    info->filterLog2BitsFinal3 = handle->filter.log2BitsF3;
    info->filterBitsSet4 = handle->filter.bitsSet4;
    info->multiProcessorCount = handle->multiProcessorCount;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFACX_getTable(PFAC_handle_t handle, PFACX_table_t which, const void **ptr, size_t *bytes)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!ptr || !bytes) return PFAC_STATUS_INVALID_PARAMETER;
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    *ptr = nullptr; *bytes = 0;
    switch (which) {
    case PFACX_TABLE_DENSE:
        *ptr = handle->h_dense.data(); *bytes = handle->h_dense.size() * sizeof(int); break;
    case PFACX_TABLE_HASH_ROWPTR:
        *ptr = handle->h_hashRow.data(); *bytes = handle->h_hashRow.size() * sizeof(Int2); break;
    case PFACX_TABLE_HASH_VALPTR:
        *ptr = handle->h_hashVal.data(); *bytes = handle->h_hashVal.size() * sizeof(Int2); break;
    case PFACX_TABLE_INITIAL_ROW:
        *ptr = handle->h_initialRow.data(); *bytes = handle->h_initialRow.size() * sizeof(int); break;
    case PFACX_TABLE_FILTER_GRAM3:
        *ptr = handle->filter.gram3.data(); *bytes = handle->filter.gram3.size() * sizeof(uint32_t); break;
    case PFACX_TABLE_FILTER_SHORT:
        *ptr = handle->filter.shortBits.data(); *bytes = handle->filter.shortBits.size() * sizeof(uint32_t); break;
    case PFACX_TABLE_FILTER_GRAM4:
        *ptr = handle->filter.gram4.data(); *bytes = handle->filter.gram4.size() * sizeof(uint32_t); break;
    case PFACX_TABLE_FILTER_FINAL3:
        *ptr = handle->filter.final3.data(); *bytes = handle->filter.final3.size() * sizeof(uint32_t); break;
    default: return PFAC_STATUS_INVALID_PARAMETER;
    }
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFACX_getScanStats(PFAC_handle_t handle, PFACX_scan_stats_t *stats)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!stats) return PFAC_STATUS_INVALID_PARAMETER;
    std::memset(stats, 0, sizeof(*stats));
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    if (!handle->hasDevice || !handle->d_workCounters) return PFAC_STATUS_LIB_NOT_EXIST;
    unsigned long long v[pfac::kStatsCount];
    if (hipStreamSynchronize(nullptr) != hipSuccess ||
        hipMemcpy(v, handle->d_workCounters + pfac::kStatsWord, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
        return PFAC_STATUS_INTERNAL_ERROR;
    stats->walkerRounds = v[0]; stats->laneSteps = v[1]; stats->walksStarted = v[2]; stats->level1Hits = v[3];
    stats->tilesPerChunk = pfac::kChunkTiles;
    stats->walksPerLane = PFAC_WALK_SETS;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFACX_setKernelVariant(PFAC_handle_t handle, int variant)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (variant != PFACX_KERNEL_FILTER && variant != PFACX_KERNEL_NAIVE && variant != PFACX_KERNEL_AUTO) return PFAC_STATUS_INVALID_PARAMETER;
    handle->kernelVariant = variant;
    return PFAC_STATUS_SUCCESS;
}

} /* extern "C" */

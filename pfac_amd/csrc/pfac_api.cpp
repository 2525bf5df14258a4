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
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

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
    std::vector<pfac::ChainSlot>().swap(c->h_chainSlots);
    c->numChainSlots = 0;
    c->chainJumpLog2 = 0;
    c->numOfTableEntry = c->sizeOfTableEntry = c->sizeOfTableInBytes = 0;
}

void freeHostStage(PFAC_context *c)
{
    for (int b = 0; b < 2; b++) {
        devFree(c->d_stageIn[b]);
        devFree(c->d_stageOut[b]);
        devFree(c->d_stagePos[b]);
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
    devFree(c->d_ladder);
    devFree(c->d_gram1);
    devFree(c->d_prefix4);
    devFree(c->d_workCounters);
    if (c->h_modeHint) { (void)hipHostFree(c->h_modeHint); c->h_modeHint = c->d_modeHint = nullptr; }
    devFree(c->d_reduceScratch);
    c->reduceScratchBytes = 0;
    freeHostStage(c);
    for (auto &e : c->evTime) { if (e) (void)hipEventDestroy(static_cast<hipEvent_t>(e)); e = nullptr; }
    c->kernelTiming = c->evTimeRecorded = false;
    devFree(c->d_final3);
    devFree(c->d_denseList);
    c->denseListEntries = 0;
    for (auto &child : c->children) (void)PFAC_destroy(child.second);
    c->children.clear();
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

/* Build (unless a compiled file brought it along) and upload the chained device table (tables.cpp:
 * buildChainedHashTable): what the scan kernel walks in BOTH perf modes.  It is built from the trie; the handle's
 * reference-layout table (PFACX_getTable, the dump, the simple kernel) is dense or hashed as the perf mode says. */
PFAC_status_t uploadChainedHashTable(PFAC_context *c)
{
    PFAC_status_t st = PFAC_STATUS_SUCCESS;
    if (c->h_chainSlots.empty()) {
        st = pfac::buildChainedHashTable(c->fa, c->h_chainSlots, c->chainJumpLog2);
        if (st != PFAC_STATUS_SUCCESS) return st;
    }
    if (!c->hasDevice) return PFAC_STATUS_SUCCESS;
    c->numChainSlots = c->h_chainSlots.size();
    st = upload(c->d_chainSlots, c->h_chainSlots.data(), c->h_chainSlots.size());
    return st;
}

/* The reference-layout table of the perf mode on the HOST: the dense table is materialised on first use -- PFACX_getTable,
 * the CPU platforms, PFACX_KERNEL_REFTABLE -- because neither GPU kernel of the product path reads it (both walk the
 * chained table) and it is S KiB: 498 MB for a Snort-scale set.  The hashed tables (a few MB) are built with the set. */
PFAC_status_t ensureHostRefTable(PFAC_context *c, bool tablesLocked = false)
{
    if (c->perfMode == PFAC_TIME_DRIVEN && c->h_dense.empty()) {
        if (tablesLocked) return pfac::buildDenseTable(c->fa, c->h_dense);
        std::unique_lock<std::shared_mutex> w(c->tablesInUse);         /* (the caller holds c->lock) */
        return pfac::buildDenseTable(c->fa, c->h_dense);
    }
    return PFAC_STATUS_SUCCESS;
}

/* ... and on the DEVICE: only the reference-shaped kernel (PFACX_KERNEL_REFTABLE) reads it there */
PFAC_status_t ensureDeviceRefTable(PFAC_context *c, bool tablesLocked = false)
{
    if (!c->hasDevice) return PFAC_STATUS_SUCCESS;
    PFAC_status_t st = ensureHostRefTable(c, tablesLocked);
    if (st != PFAC_STATUS_SUCCESS) return st;
    if (c->perfMode == PFAC_TIME_DRIVEN) {
        if (!c->d_dense) st = upload(c->d_dense, c->h_dense.data(), c->h_dense.size());
    } else if (!c->d_hashRow || !c->d_hashVal) {
        devFree(c->d_hashRow);
        devFree(c->d_hashVal);
        st = upload(c->d_hashRow, c->h_hashRow.data(), c->h_hashRow.size());
        if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_hashVal, c->h_hashVal.data(), c->h_hashVal.size());
    }
    return st;
}

/* ref PFAC_bindTable -> PFAC_create2DTable / PFAC_createHashTable, PFAC.cpp:321-648 */
PFAC_status_t bindTable(PFAC_context *c)
{
    if (!c->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    PFAC_status_t st;
    if (c->perfMode == PFAC_TIME_DRIVEN) {
        c->numOfTableEntry = (size_t)pfac::kCharSet * (size_t)c->fa.numStates;
        c->sizeOfTableEntry = sizeof(int);
    } else {
        if (c->h_hashRow.empty()) {
            st = pfac::buildHashTable(c->fa, c->h_hashRow, c->h_hashVal);
            if (st != PFAC_STATUS_SUCCESS) return st;
        }
        c->numOfTableEntry = c->h_hashVal.size();
        c->sizeOfTableEntry = sizeof(Int2);
    }
    c->sizeOfTableInBytes = c->numOfTableEntry * c->sizeOfTableEntry;
    if (c->hasDevice && !c->d_chainSlots) {
        st = uploadChainedHashTable(c);
        if (st == PFAC_STATUS_SUCCESS && c->kernelVariant == PFACX_KERNEL_REFTABLE) st = ensureDeviceRefTable(c, /*tablesLocked=*/true);   /* every caller of bindTable holds tablesInUse */
        if (st != PFAC_STATUS_SUCCESS) { freeTables(c); return st; }
    }
    return PFAC_STATUS_SUCCESS;
}

/* tables that do not depend on perfMode: initial-state row and prefilter (built, or brought along by a compiled file) */
PFAC_status_t bindCommon(PFAC_context *c, bool build = true)
{
    if (build) {
        pfac::buildInitialRow(c->fa, c->h_initialRow);
        pfac::buildFilter(c->fa, c->filter);
    }
    if (!c->hasDevice) return PFAC_STATUS_SUCCESS;
    PFAC_status_t st = upload(c->d_initialRow, c->h_initialRow.data(), c->h_initialRow.size());
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_gram3, c->filter.gram3.data(), c->filter.gram3.size());
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_shortBits, c->filter.shortBits.data(), c->filter.shortBits.size());
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_ladder, c->filter.ladder.data(), c->filter.ladder.size());
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_final3, c->filter.final3.data(), c->filter.final3.size());
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_gram1, c->filter.gram1.data(), c->filter.gram1.size());
    if (st == PFAC_STATUS_SUCCESS) st = upload(c->d_prefix4, c->filter.prefix4.data(), c->filter.prefix4.size());
    if (st == PFAC_STATUS_SUCCESS) {               /* chunk counters of the scan kernel, reset before every launch */
        const std::vector<unsigned int> zeros(pfac::kWorkCounterWords, 0u);
        st = upload(c->d_workCounters, zeros.data(), zeros.size());
    }
    if (st == PFAC_STATUS_SUCCESS && !c->h_modeHint) {
        /* the word the scan kernel tells the host through what the stream looked like (pfac_context.h); without it AUTO means
         * the register-window walker */
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            c->h_modeHint = static_cast<unsigned int *>(h);
            c->d_modeHint = static_cast<unsigned int *>(d);
            *c->h_modeHint = 0;
        } else {
            if (h) (void)hipHostFree(h);
            (void)hipGetLastError();
        }
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
    {   /* the dense table is built on first use (ensureHostRefTable) */
        std::lock_guard<std::mutex> guard(c->lock);
        const PFAC_status_t st = ensureHostRefTable(c);
        if (st != PFAC_STATUS_SUCCESS) return st;
    }
    bool omp = false;
    if (c->platform == PFAC_PLATFORM_CPU_OMP) omp = (std::getenv("OMP_NUM_THREADS") != nullptr);
    std::shared_lock<std::shared_mutex> r(c->tablesInUse);             /* a setter on another thread waits until the match is through */
    if (c->perfMode == PFAC_TIME_DRIVEN && c->h_dense.empty()) return PFAC_STATUS_PATTERNS_NOT_READY;   /* ... or has just replaced the set: its tables are built on the next call */
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
    /* drops this handle's reference; the module stays mapped while other handles hold theirs (dlopen refcounts) */
    if (handle->module) dlclose(handle->module);
    delete handle;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFAC_setPlatform(PFAC_handle_t handle, PFAC_platform_t platform)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (platform != PFAC_PLATFORM_GPU && platform != PFAC_PLATFORM_CPU && platform != PFAC_PLATFORM_CPU_OMP)
        return PFAC_STATUS_INVALID_PARAMETER;
    std::lock_guard<std::mutex> guard(handle->lock);
    handle->platform = (int)platform;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFAC_setTextureMode(PFAC_handle_t handle, PFAC_textureMode_t textureModeSel)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (textureModeSel != PFAC_AUTOMATIC && textureModeSel != PFAC_TEXTURE_ON && textureModeSel != PFAC_TEXTURE_OFF)
        return PFAC_STATUS_INVALID_PARAMETER;
    std::lock_guard<std::mutex> guard(handle->lock);
    handle->textureMode = (int)textureModeSel;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFAC_setPerfMode(PFAC_handle_t handle, PFAC_perfMode_t perfModeSel)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (perfModeSel != PFAC_TIME_DRIVEN && perfModeSel != PFAC_SPACE_DRIVEN) return PFAC_STATUS_INVALID_PARAMETER;
    std::lock_guard<std::mutex> guard(handle->lock);
    std::unique_lock<std::shared_mutex> tables(handle->tablesInUse);
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
    std::lock_guard<std::mutex> guard(handle->lock);
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

static PFAC_status_t readFromFile(PFAC_handle_t handle, const char *filename, unsigned int flags)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!filename || (flags & ~(PFACX_READ_STRICT | PFACX_READ_STRIP_CR))) return PFAC_STATUS_INVALID_PARAMETER;
    std::lock_guard<std::mutex> guard(handle->lock);
    std::unique_lock<std::shared_mutex> tables(handle->tablesInUse);
    if (handle->isPatternsReady) freeResources(handle);            /* ref PFAC.cpp:663-666 */
    if (std::strlen(filename) >= (size_t)pfac::kFileNameLen) return PFAC_STATUS_INTERNAL_ERROR;  /* ref :668-672 */
    handle->patternFile = filename;

    PFAC_status_t st = pfac::compilePatternFile(filename, handle->fa, flags);
    if (st != PFAC_STATUS_SUCCESS) { freeResources(handle); return st; }
    handle->isPatternsReady = true;
    st = bindCommon(handle);
    if (st == PFAC_STATUS_SUCCESS) st = bindTable(handle);
    if (st != PFAC_STATUS_SUCCESS) { freeResources(handle); return st; }
    return PFAC_STATUS_SUCCESS;
}

static PFAC_status_t readFromMemory(PFAC_handle_t handle, const char *patterns, size_t size, unsigned int flags)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if ((!patterns && size) || (flags & ~(PFACX_READ_STRICT | PFACX_READ_STRIP_CR))) return PFAC_STATUS_INVALID_PARAMETER;
    std::lock_guard<std::mutex> guard(handle->lock);
    std::unique_lock<std::shared_mutex> tables(handle->tablesInUse);
    if (handle->isPatternsReady) freeResources(handle);
    handle->patternFile.clear();
    PFAC_status_t st;
    try {
        st = pfac::compilePatternBytes(std::vector<unsigned char>(patterns, patterns + size), handle->fa, flags);
    } catch (const std::bad_alloc &) { st = PFAC_STATUS_ALLOC_FAILED; }
    if (st != PFAC_STATUS_SUCCESS) { freeResources(handle); return st; }
    handle->isPatternsReady = true;
    st = bindCommon(handle);
    if (st == PFAC_STATUS_SUCCESS) st = bindTable(handle);
    if (st != PFAC_STATUS_SUCCESS) { freeResources(handle); return st; }
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFAC_readPatternFromFile(PFAC_handle_t handle, char *filename) { return readFromFile(handle, filename, 0); }

/* pfac_ext.h: the same pattern-file bytes from memory instead of from a file */
PFAC_status_t PFACX_readPatternFromMemory(PFAC_handle_t handle, const char *patterns, size_t size) { return readFromMemory(handle, patterns, size, 0); }

/* pfac_ext.h: ... with options: strict about a last line without a newline, CRLF line ends */
PFAC_status_t PFACX_readPatternFromFileEx(PFAC_handle_t handle, const char *filename, unsigned int flags) { return readFromFile(handle, filename, flags); }
PFAC_status_t PFACX_readPatternFromMemoryEx(PFAC_handle_t handle, const char *patterns, size_t size, unsigned int flags)
{
    return readFromMemory(handle, patterns, size, flags);
}

} /* extern "C" */

namespace {

/* PFAC_matchFromDevice behind the argument checks; the caller holds handle->lock */
PFAC_status_t matchDeviceLocked(PFAC_context *c, char *d_inputString, size_t size, int *d_matched_result)
{
    if (!c->hasDevice || !c->module) return PFAC_STATUS_LIB_NOT_EXIST;      /* never a CPU fallback */
    correctTextureMode(c);
    if (c->perfMode == PFAC_TIME_DRIVEN) return c->kernel_time_driven_ptr(c, d_inputString, size, d_matched_result);
    if (c->perfMode == PFAC_SPACE_DRIVEN) return c->kernel_space_driven_ptr(c, d_inputString, size, d_matched_result);
    return PFAC_STATUS_INTERNAL_ERROR;
}

/*
 * Host buffers through the GPU: results for positions [0, owned) of a stream of which `readable` >= owned bytes
 * may be read (walks that start before `owned` may run into the rest: the slices of a sharded stream,
 * omp_PFAC.cpp:324,377).  The caller holds c->lock.
 *
 * The reference allocates, uploads, scans, downloads and frees in sequence (PFAC.cpp:916-960), which leaves the
 * scan idle for the 5 bytes per position that cross the host link.  Here the stream is cut into pieces of
 * kHostPiece positions: piece i+1 is uploaded and piece i-1 downloaded while piece i is scanned (SURVEY 8f
 * rank 2).  Each piece is scanned together with the maxPatternLen bytes behind it -- a walk may read that far --
 * and only its own results go back.  The staging buffers, two copy streams and their events belong to the
 * handle and are created on first use; the scan itself stays on the default stream.
 */
static PFAC_status_t ensureHostStage(PFAC_context *c, size_t need)
{
    if (c->hostStagePositions >= need) return PFAC_STATUS_SUCCESS;
    freeHostStage(c);
    bool ok = true;
    for (int b = 0; b < 2 && ok; b++) {
        ok = hipMalloc(reinterpret_cast<void **>(&c->d_stageIn[b]), (need + 3) & ~size_t(3)) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&c->d_stageOut[b]), need * sizeof(int)) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&c->d_stagePos[b]), need * sizeof(int)) == hipSuccess;
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
    return PFAC_STATUS_SUCCESS;
}

/* every result crosses the link: pieces with many matches */
static PFAC_status_t matchHostFullVector(PFAC_context *c, char *h_inputString, size_t owned, size_t readable, int *h_matched_result)
{
    const size_t overlap = (size_t)c->fa.maxPatternLen;
    const size_t piece = owned < kHostPiece ? owned : kHostPiece;
    PFAC_status_t st = ensureHostStage(c, piece + overlap);
    if (st != PFAC_STATUS_SUCCESS) return st;
    hipStream_t up = static_cast<hipStream_t>(c->stageUp), down = static_cast<hipStream_t>(c->stageDown);
    bool used[2] = {false, false};
    size_t i = 0;
    for (size_t off = 0; off < owned && st == PFAC_STATUS_SUCCESS; off += piece, i++) {
        const int b = (int)(i & 1);
        const size_t mine = owned - off < piece ? owned - off : piece;
        const size_t scanned = readable - off < mine + overlap ? readable - off : mine + overlap;
        hipEvent_t evUp = static_cast<hipEvent_t>(c->evUp[b]), evScan = static_cast<hipEvent_t>(c->evScan[b]),
                   evDown = static_cast<hipEvent_t>(c->evDown[b]);
        bool ok = true;
        if (used[b]) ok = hipStreamWaitEvent(up, evScan, 0) == hipSuccess;          /* the scan of piece i-2 has read this buffer */
        ok = ok && hipMemcpyAsync(c->d_stageIn[b], h_inputString + off, scanned, hipMemcpyHostToDevice, up) == hipSuccess &&
             hipEventRecord(evUp, up) == hipSuccess && hipStreamWaitEvent(nullptr, evUp, 0) == hipSuccess;
        if (ok && used[b]) ok = hipStreamWaitEvent(nullptr, evDown, 0) == hipSuccess;   /* its results have left this buffer */
        if (!ok) { st = PFAC_STATUS_INTERNAL_ERROR; break; }
        st = matchDeviceLocked(c, c->d_stageIn[b], scanned, c->d_stageOut[b]);
        if (st != PFAC_STATUS_SUCCESS) break;
        ok = hipEventRecord(evScan, nullptr) == hipSuccess && hipStreamWaitEvent(down, evScan, 0) == hipSuccess &&
             hipMemcpyAsync(h_matched_result + off, c->d_stageOut[b], mine * sizeof(int), hipMemcpyDeviceToHost, down) == hipSuccess &&
             hipEventRecord(evDown, down) == hipSuccess;
        if (!ok) st = PFAC_STATUS_INTERNAL_ERROR;
        used[b] = true;
    }
    const bool drained = hipStreamSynchronize(up) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess &&
                         hipStreamSynchronize(down) == hipSuccess;
    if (!drained && st == PFAC_STATUS_SUCCESS) st = PFAC_STATUS_INTERNAL_ERROR;
    return st;
}

/*
 * PFAC_matchFromHost on the GPU.  Four of the five bytes per position that the reference moves over the host link
 * (PFAC.cpp:916-960) are results, and nearly all of them are zero.  So the pieces are scanned with the compacted-
 * output kernel and only the (position, id) pairs come back; the zeros are written where they are needed -- by a few
 * helper threads of this call straight into the caller's result vector, while the pieces are uploaded and scanned --
 * and the pairs are scattered on top at the end.  A piece in which more than one position in eight matches takes the
 * full-vector route above instead (after the zero fill, so the two never write the same words at the same time).
 */
/* The NUMA node a host page lives on (-1: unknown, not faulted in yet, or no such system call): move_pages with no target only reports. */
static int numaNodeOf(const void *p)
{
#if defined(__linux__) && defined(SYS_move_pages)
    void *page = reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(4095));
    int status = -1;
    if (syscall(SYS_move_pages, 0, 1UL, &page, nullptr, &status, 0) == 0 && status >= 0) return status;
#else
    (void)p;
#endif
    return -1;
}
/* the CPUs of a NUMA node that this thread may run on (empty: unknown) */
static bool cpusOfNumaNode(int node, cpu_set_t &out)
{
    CPU_ZERO(&out);
    char path[96];
    std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = std::fopen(path, "r");
    if (!f) return false;
    char buf[4096];
    const size_t got = std::fread(buf, 1, sizeof(buf) - 1, f);
    std::fclose(f);
    buf[got] = 0;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    int any = 0;
    for (char *q = buf; *q;) {
        char *end = nullptr;
        const long a = std::strtol(q, &end, 10);
        if (end == q) break;
        long b = a;
        if (*end == '-') { q = end + 1; b = std::strtol(q, &end, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, &out); any++; }
        q = (*end == ',') ? end + 1 : end;
        if (*end != ',' ) break;
    }
    return any > 0;
}

/* zeros without reading the lines first: streaming stores, 64 bytes per trip (the result vector of a 1 GiB call is 4 GiB
 * that nothing reads before the caller does) */
static void fillZeroStreaming(int *p, size_t n)
{
#if !defined(__SSE2__)
    std::memset(p, 0, n * sizeof(int));                        /* hosts without SSE2 (aarch64, ppc64 nodes with AMD GPUs): plain stores */
    return;
#else
    static const bool plain = std::getenv("PFAC_HOST_FILL_MEMSET") != nullptr;
    if (plain) { std::memset(p, 0, n * sizeof(int)); return; }
    while (n && (reinterpret_cast<uintptr_t>(p) & 63u)) { *p++ = 0; n--; }
    const __m128i z = _mm_setzero_si128();
    for (; n >= 16; n -= 16, p += 16) {
        _mm_stream_si128(reinterpret_cast<__m128i *>(p), z);
        _mm_stream_si128(reinterpret_cast<__m128i *>(p + 4), z);
        _mm_stream_si128(reinterpret_cast<__m128i *>(p + 8), z);
        _mm_stream_si128(reinterpret_cast<__m128i *>(p + 12), z);
    }
    while (n) { *p++ = 0; n--; }
    _mm_sfence();
#endif
}

PFAC_status_t matchHostOnGpu(PFAC_context *c, char *h_inputString, size_t owned, size_t readable, int *h_matched_result)
{
    if (!c->hasDevice || !c->module) return PFAC_STATUS_LIB_NOT_EXIST;
    const size_t overlap = (size_t)c->fa.maxPatternLen;
    const size_t piece = owned < kHostPiece ? owned : kHostPiece;
    PFAC_status_t st = ensureHostStage(c, piece + overlap);
    if (st != PFAC_STATUS_SUCCESS) return st;
    correctTextureMode(c);
    PFAC_reduce_kernel_protoType reduce = c->perfMode == PFAC_TIME_DRIVEN ? c->reduce_kernel_ptr : c->reduce_inplace_kernel_ptr;
    hipStream_t up = static_cast<hipStream_t>(c->stageUp);
    const size_t numPieces = (owned + piece - 1) / piece;
    auto uploadPiece = [&](size_t i) -> bool {               /* into buffer i & 1, on the upload stream */
        const size_t off = i * piece;
        const size_t mine = owned - off < piece ? owned - off : piece;
        const size_t scanned = readable - off < mine + overlap ? readable - off : mine + overlap;
        return hipMemcpyAsync(c->d_stageIn[i & 1], h_inputString + off, scanned, hipMemcpyHostToDevice, up) == hipSuccess &&
               hipEventRecord(static_cast<hipEvent_t>(c->evUp[i & 1]), up) == hipSuccess;
    };
    /* the link first: nothing below is worth a microsecond of an idle copy engine */
    const bool trace = std::getenv("PFAC_HOST_TRACE") != nullptr;
    const auto tStart = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tStart).count(); };
    /* The uploads are queued by a thread of their own: hipMemcpyAsync from PAGEABLE memory does not return until the runtime
     * has staged the piece (0.6 ms for 32 MiB), and this thread has the scans to launch and their pairs to fetch meanwhile.
     * Piece i goes into buffer i & 1 once the scan of piece i - 2 is over. */
    std::atomic<size_t> scansDone{0}, uploadsQueued{0};
    std::atomic<bool> uploadFailed{false}, stopUploads{false};
    int device = 0;
    (void)hipGetDevice(&device);
    std::thread uploader;
    bool ok = true;
    if (numPieces == 1) {                                      /* nothing to overlap with: no thread (tens of microseconds of a small call) */
        ok = uploadPiece(0);
        uploadsQueued.store(1);
    } else {
        try {
            uploader = std::thread([&]() {
                if (hipSetDevice(device) != hipSuccess) { uploadFailed.store(true); return; }
                for (size_t i = 0; i < numPieces; i++) {
                    while (i >= 2 && scansDone.load(std::memory_order_acquire) + 1 < i && !stopUploads.load(std::memory_order_relaxed)) std::this_thread::yield();
                    if (stopUploads.load(std::memory_order_relaxed)) return;
                    if (!uploadPiece(i)) { uploadFailed.store(true); return; }
                    uploadsQueued.store(i + 1, std::memory_order_release);
                }
            });
        } catch (...) { ok = false; }
    }
    const double tUp0 = since();

    /* Zero fill of the caller's vector, in parallel with everything below: 4 bytes of host memory per position against 1 byte
     * over the link, so it takes a few threads -- sized from the cores this thread may run on (a caller bound to a cpuset has
     * fewer than the machine) up to 8: the fill and the link's reads share the host's memory channels, and beyond eight
     * threads the upload loses more than the fill gains (256 MiB from pinned buffers on a 2 x 64-core box, link 54 GB/s:
     * 47.0 / 48.8 / 43.8 / 43.3 / 46.4 GB/s with 4 / 8 / 12 / 16 / 24 threads; memset instead of streaming stores: 24.7) --
     * streaming stores, and the pieces IN ORDER, every thread its share of each: the pairs of piece k are scattered as soon
     * as they are back, while piece k + 1 uploads, not in one pass at the end.  (PFAC_HOST_FILL_THREADS overrides the count:
     * a measurement aid.) */
    unsigned helpers = 0;
    if (owned >= (size_t(4) << 20)) {
        unsigned hw = std::thread::hardware_concurrency();
        cpu_set_t allowed;
        if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) hw = (unsigned)CPU_COUNT(&allowed);
        helpers = hw >= 64 ? 8 : hw >= 16 ? 4 : hw >= 4 ? 2 : 1;
        if (const char *e = std::getenv("PFAC_HOST_FILL_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 256) helpers = (unsigned)v; }
    }
    auto share = [&](size_t k, unsigned t, unsigned of, size_t &lo, size_t &hi) {          /* thread t's part of piece k */
        const size_t off = k * piece, mine = owned - off < piece ? owned - off : piece;
        lo = off + mine * t / of / 16 * 16;
        hi = t + 1 == of ? off + mine : off + mine * (t + 1) / of / 16 * 16;
    };
    std::unique_ptr<std::atomic<unsigned>[]> filled;
    std::vector<std::thread> fillers;
    try {
        filled.reset(new std::atomic<unsigned>[numPieces]);
        for (size_t k = 0; k < numPieces; k++) filled[k].store(0, std::memory_order_relaxed);
        fillers.reserve(helpers);
        /* The fill threads run on the NUMA node the caller's result vector lives on: 4 bytes per position of streaming stores that
         * cross the sockets' link meet the link's own reads of the input there (2 x EPYC 9575F, GPU on node 0, pinned buffers
         * first-touched on node 1: p50 7.4 ms, p90 11.4 ms per 256 MiB call against 5.5 / 6.2 ms with the buffers on node 0 --
         * the driver's round-4 line: 29 GB/s median; tools/host_numa_probe.py).  PFAC_HOST_FILL_ANYWHERE=1 leaves them to the OS. */
        cpu_set_t fillCpus;
        bool bindFill = false;
        if (helpers && std::getenv("PFAC_HOST_FILL_ANYWHERE") == nullptr) {
            const int node = numaNodeOf(h_matched_result + owned / 2);
            bindFill = node >= 0 && cpusOfNumaNode(node, fillCpus);
        }
        for (unsigned t = 0; t < helpers; t++)
            fillers.emplace_back([&, t]() {
                if (bindFill) (void)pthread_setaffinity_np(pthread_self(), sizeof(fillCpus), &fillCpus);
                for (size_t k = 0; k < numPieces; k++) {
                    size_t lo, hi;
                    share(k, t, helpers, lo, hi);
                    fillZeroStreaming(h_matched_result + lo, hi - lo);
                    filled[k].fetch_add(1, std::memory_order_release);
                }
            });
    } catch (...) { /* no memory, or fewer threads than planned: the shares nobody started are filled by this thread, below */ }
    if (!filled) {                                             /* not even the counters: no helper was started */
        std::memset(h_matched_result, 0, owned * sizeof(int));
        helpers = 0;
    }
    const unsigned started = (unsigned)fillers.size();
    const double tThreads = since();
    auto joinAll = [&]() { for (std::thread &t : fillers) if (t.joinable()) t.join(); };
    /* piece k of the caller's vector is all zeros when this returns */
    auto waitFilled = [&](size_t k) {
        if (!filled) return;
        if (helpers == 0) {                                    /* a small call: this thread fills, piece by piece */
            size_t lo, hi;
            share(k, 0, 1, lo, hi);
            if (filled[k].load(std::memory_order_relaxed) == 0) { std::memset(h_matched_result + lo, 0, (hi - lo) * sizeof(int)); filled[k].store(1, std::memory_order_relaxed); }
            return;
        }
        if (filled[k].load(std::memory_order_acquire) < helpers) {          /* acquire: the pairs are scattered onto words the fillers wrote */
            for (unsigned t = started; t < helpers; t++) {          /* the shares of threads that could not be started */
                size_t lo, hi;
                share(k, t, helpers, lo, hi);
                fillZeroStreaming(h_matched_result + lo, hi - lo);
            }
            while (filled[k].load(std::memory_order_acquire) < started) std::this_thread::yield();
            filled[k].store(helpers, std::memory_order_relaxed);
        }
    };

    std::vector<int> pos, id;                                  /* the pairs of one piece */
    std::vector<size_t> densePieces;
    try {
        for (size_t i = 0; i < numPieces && ok && st == PFAC_STATUS_SUCCESS; i++) {
            const int b = (int)(i & 1);
            const size_t off = i * piece;
            const size_t mine = owned - off < piece ? owned - off : piece;
            const size_t scanned = readable - off < mine + overlap ? readable - off : mine + overlap;
            while (uploadsQueued.load(std::memory_order_acquire) <= i && !uploadFailed.load(std::memory_order_relaxed)) std::this_thread::yield();
            ok = !uploadFailed.load(std::memory_order_relaxed) && hipStreamWaitEvent(nullptr, static_cast<hipEvent_t>(c->evUp[b]), 0) == hipSuccess;
            if (!ok) break;
            int count = 0;
            c->reduceUnordered = true;
            st = reduce(c, reinterpret_cast<int *>(c->d_stageIn[b]), (int)scanned, c->d_stageOut[b], c->d_stagePos[b], &count, nullptr, nullptr);
            c->reduceUnordered = false;
            if (st != PFAC_STATUS_SUCCESS) break;
            scansDone.store(i + 1, std::memory_order_release);     /* the scan is synchronous: its input buffer may take piece i + 2 */
            if ((size_t)count > mine / 8) { densePieces.push_back(i); continue; }
            pos.resize((size_t)count);
            id.resize((size_t)count);
            if (count && (hipMemcpy(pos.data(), c->d_stagePos[b], (size_t)count * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
                          hipMemcpy(id.data(), c->d_stageOut[b], (size_t)count * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)) {
                ok = false;
                break;
            }
            waitFilled(i);                                     /* long done, as a rule: the fill runs ahead of the link */
            for (size_t k = 0; k < pos.size(); k++)
                if ((size_t)pos[k] < mine) h_matched_result[off + (size_t)pos[k]] = id[k];   /* beyond: the next piece's (or nobody's) */
        }
        if (!ok && st == PFAC_STATUS_SUCCESS) st = PFAC_STATUS_INTERNAL_ERROR;
    } catch (const std::bad_alloc &) { st = PFAC_STATUS_ALLOC_FAILED; }
    stopUploads.store(true);
    if (uploader.joinable()) uploader.join();
    const double tLoop = since();
    const bool drained = hipStreamSynchronize(up) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess;
    if (!drained && st == PFAC_STATUS_SUCCESS) st = PFAC_STATUS_INTERNAL_ERROR;
    const double tDrained = since();
    for (size_t k = 0; k < numPieces; k++) waitFilled(k);      /* every element of the caller's vector is written, whatever happened */
    joinAll();
    if (trace) std::fprintf(stderr, "PFAC_HOST_TRACE %zu B %zu pieces %u helpers: first upload queued %.3f ms, threads started %.3f, piece loop done %.3f, drained %.3f, filled+joined %.3f\n",
                            owned, numPieces, started, tUp0, tThreads, tLoop, tDrained, since());
    if (st != PFAC_STATUS_SUCCESS) return st;
    for (size_t i : densePieces) {
        const size_t off = i * piece;
        const size_t mine = owned - off < piece ? owned - off : piece;
        st = matchHostFullVector(c, h_inputString + off, mine, readable - off, h_matched_result + off);
        if (st != PFAC_STATUS_SUCCESS) return st;
    }
    return PFAC_STATUS_SUCCESS;
}

/*
 * PFAC_matchFromHostReduce on the GPU (ref PFAC.cpp:1010-1128: one allocation of size + 8 * size device bytes, one blocking
 * copy, one scan, two copies back).  Same pipeline as PFAC_matchFromHost: the stream goes through the handle's staging
 * buffers in pieces of kHostReducePiece positions, piece i + 1 is uploaded (by a thread of its own: see matchHostOnGpu) while
 * piece i is scanned by the compacted-output kernel -- together with the maxPatternLen bytes behind it -- and its pairs, in
 * position order, are copied straight behind those of the pieces before it: pieces are in stream order, so the whole list
 * is.  A pair whose position lies in the overlap belongs to the next piece, which finds it again.  Device memory: two
 * pieces (9 bytes per position) instead of 9 bytes for every position of the stream.
 */
constexpr size_t kHostReducePiece = size_t(16) << 20;
PFAC_status_t matchHostReduceOnGpu(PFAC_context *c, char *h_inputString, size_t size, int *h_matched_result, int *h_pos, int *h_num_matched)
{
    if (!c->hasDevice || !c->module) return PFAC_STATUS_LIB_NOT_EXIST;
    const size_t overlap = (size_t)c->fa.maxPatternLen;
    const size_t piece = size < kHostReducePiece ? size : kHostReducePiece;
    PFAC_status_t st = ensureHostStage(c, piece + overlap);
    if (st != PFAC_STATUS_SUCCESS) return st;
    correctTextureMode(c);
    PFAC_reduce_kernel_protoType reduce = c->perfMode == PFAC_TIME_DRIVEN ? c->reduce_kernel_ptr : c->reduce_inplace_kernel_ptr;
    hipStream_t up = static_cast<hipStream_t>(c->stageUp);
    const size_t numPieces = (size + piece - 1) / piece;
    auto uploadPiece = [&](size_t i) -> bool {               /* into buffer i & 1, on the upload stream */
        const size_t off = i * piece;
        const size_t mine = size - off < piece ? size - off : piece;
        const size_t scanned = size - off < mine + overlap ? size - off : mine + overlap;
        return hipMemcpyAsync(c->d_stageIn[i & 1], h_inputString + off, scanned, hipMemcpyHostToDevice, up) == hipSuccess &&
               hipEventRecord(static_cast<hipEvent_t>(c->evUp[i & 1]), up) == hipSuccess;
    };
    std::atomic<size_t> scansDone{0}, uploadsQueued{0};
    std::atomic<bool> uploadFailed{false}, stopUploads{false};
    int device = 0;
    (void)hipGetDevice(&device);
    std::thread uploader;
    bool ok = true;
    if (numPieces == 1) {                                      /* nothing to overlap: no thread */
        ok = uploadPiece(0);
        uploadsQueued.store(1);
    } else {
        try {
            uploader = std::thread([&]() {
                if (hipSetDevice(device) != hipSuccess) { uploadFailed.store(true); return; }
                for (size_t i = 0; i < numPieces; i++) {
                    while (i >= 2 && scansDone.load(std::memory_order_acquire) + 1 < i && !stopUploads.load(std::memory_order_relaxed)) std::this_thread::yield();
                    if (stopUploads.load(std::memory_order_relaxed)) return;
                    if (!uploadPiece(i)) { uploadFailed.store(true); return; }
                    uploadsQueued.store(i + 1, std::memory_order_release);
                }
            });
        } catch (...) { ok = false; }
    }
    size_t total = 0;
    for (size_t i = 0; i < numPieces && ok && st == PFAC_STATUS_SUCCESS; i++) {
        const int b = (int)(i & 1);
        const size_t off = i * piece;
        const size_t mine = size - off < piece ? size - off : piece;
        const size_t scanned = size - off < mine + overlap ? size - off : mine + overlap;
        while (uploadsQueued.load(std::memory_order_acquire) <= i && !uploadFailed.load(std::memory_order_relaxed)) std::this_thread::yield();
        ok = !uploadFailed.load(std::memory_order_relaxed) && hipStreamWaitEvent(nullptr, static_cast<hipEvent_t>(c->evUp[b]), 0) == hipSuccess;
        if (!ok) break;
        int count = 0;
        st = reduce(c, reinterpret_cast<int *>(c->d_stageIn[b]), (int)scanned, c->d_stageOut[b], c->d_stagePos[b], &count, nullptr, nullptr);
        if (st != PFAC_STATUS_SUCCESS) break;
        scansDone.store(i + 1, std::memory_order_release);     /* the scan is synchronous: its input buffer may take piece i + 2 */
        if (count == 0) continue;
        /* total <= off (a position has at most one pair) and count <= scanned <= size - off: the caller's arrays (size entries) hold them */
        if (hipMemcpy(h_pos + total, c->d_stagePos[b], (size_t)count * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        size_t keep = (size_t)count;                           /* positions ascend: those in the overlap are a suffix */
        while (keep > 0 && (size_t)h_pos[total + keep - 1] >= mine) keep--;
        if (keep && hipMemcpy(h_matched_result + total, c->d_stageOut[b], keep * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        if (off) for (size_t k = 0; k < keep; k++) h_pos[total + k] += (int)off;
        total += keep;
    }
    if (!ok && st == PFAC_STATUS_SUCCESS) st = PFAC_STATUS_INTERNAL_ERROR;
    stopUploads.store(true);
    if (uploader.joinable()) uploader.join();
    const bool drained = hipStreamSynchronize(up) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess;
    if (!drained && st == PFAC_STATUS_SUCCESS) st = PFAC_STATUS_INTERNAL_ERROR;
    if (st == PFAC_STATUS_SUCCESS) *h_num_matched = (int)total;
    return st;
}

} // namespace

extern "C" {

PFAC_status_t PFAC_matchFromDevice(PFAC_handle_t handle, char *d_inputString, size_t size, int *d_matched_result)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;                /* check order: ref PFAC.cpp:846-861 */
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    if (!d_inputString) return PFAC_STATUS_INVALID_PARAMETER;
    if (!d_matched_result) return PFAC_STATUS_INVALID_PARAMETER;
    if (size == 0) return PFAC_STATUS_SUCCESS;
    std::lock_guard<std::mutex> guard(handle->lock);
    return matchDeviceLocked(handle, d_inputString, size, d_matched_result);
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
    std::lock_guard<std::mutex> guard(handle->lock);
    return matchHostOnGpu(handle, h_inputString, size, size, h_matched_result);
}

/*
 * pfac_ext.h: one call shards a host stream over several GPUs (what every user of the
 * reference re-writes from PFAC/test/omp_PFAC.cpp:257-394 or SimpleMultiGPU_pthread.cpp:50-174).  One worker
 * thread per listed device: hipSetDevice, a per-device handle with this handle's pattern set and modes (kept in
 * the handle for the next call), a contiguous slice of the stream scanned together with the maxPatternLen bytes
 * behind it, only the slice's own results written (omp_PFAC.cpp:324,377).  No exchange between devices.
 */
PFAC_status_t PFACX_matchFromHostMultiGPU(PFAC_handle_t handle, char *h_inputString, size_t size, int *h_matched_result,
                                          int numDevices, const int *devices)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    if (!h_inputString || !h_matched_result || numDevices < 0) return PFAC_STATUS_INVALID_PARAMETER;
    if (size == 0) return PFAC_STATUS_SUCCESS;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) { (void)hipGetLastError(); return PFAC_STATUS_LIB_NOT_EXIST; }
    std::vector<int> devs;
    if (numDevices == 0) {
        for (int d = 0; d < visible; d++) devs.push_back(d);
    } else {
        for (int i = 0; i < numDevices; i++) {
            const int d = devices ? devices[i] : i;
            if (d < 0 || d >= visible) return PFAC_STATUS_INVALID_PARAMETER;
            devs.push_back(d);
        }
    }
    std::lock_guard<std::mutex> guard(handle->lock);
    PFAC_context *c = handle;
    const size_t workers = devs.size();
    /* the i-th worker's handle: bound to devs[i]; a device listed twice gets two handles (two streams of work) */
    while (c->children.size() < workers) c->children.emplace_back(-1, nullptr);
    std::vector<PFAC_status_t> status(workers, PFAC_STATUS_SUCCESS);
    /* slice boundaries: contiguous, rounded to the 1 KiB tile (pfac_amd/sharding.py plan_slices is the Python mirror) */
    std::vector<size_t> bound(workers + 1, 0);
    for (size_t i = 1; i < workers; i++) {
        size_t b = (size * i / workers) / 1024 * 1024;
        bound[i] = b > bound[i - 1] ? b : bound[i - 1];
    }
    bound[workers] = size;
    auto work = [&](size_t i) {
        if (bound[i + 1] == bound[i]) return;
        if (hipSetDevice(devs[i]) != hipSuccess) { status[i] = PFAC_STATUS_INTERNAL_ERROR; return; }
        auto &child = c->children[i];
        if (child.second && child.first != devs[i]) { (void)PFAC_destroy(child.second); child.second = nullptr; }
        if (!child.second) {
            PFAC_handle_t h = nullptr;
            PFAC_status_t st = PFAC_create(&h);                    /* binds the current device */
            if (st == PFAC_STATUS_SUCCESS) st = PFAC_setPerfMode(h, (PFAC_perfMode_t)c->perfMode);
            if (st == PFAC_STATUS_SUCCESS) st = PFAC_setTextureMode(h, (PFAC_textureMode_t)c->textureMode);
            if (st == PFAC_STATUS_SUCCESS) st = PFACX_setKernelVariant(h, c->kernelVariant);
            if (st == PFAC_STATUS_SUCCESS)
                st = PFACX_readPatternFromMemory(h, reinterpret_cast<const char *>(c->fa.file.data()), c->fa.file.size());
            if (st != PFAC_STATUS_SUCCESS) { if (h) (void)PFAC_destroy(h); status[i] = st; return; }
            child = {devs[i], h};
        }
        PFAC_context *w = child.second;
        /* a child created by an earlier call: the parent's modes may have changed since */
        if (w->perfMode != c->perfMode) {
            const PFAC_status_t st = PFAC_setPerfMode(w, (PFAC_perfMode_t)c->perfMode);
            if (st != PFAC_STATUS_SUCCESS) { status[i] = st; return; }
        }
        if (w->kernelVariant != c->kernelVariant) {
            const PFAC_status_t st = PFACX_setKernelVariant(w, c->kernelVariant);
            if (st != PFAC_STATUS_SUCCESS) { status[i] = st; return; }
        }
        std::lock_guard<std::mutex> g(w->lock);
        w->textureMode = c->textureMode;
        status[i] = matchHostOnGpu(w, h_inputString + bound[i], bound[i + 1] - bound[i], size - bound[i], h_matched_result + bound[i]);
    };
    int callerDevice = 0;
    (void)hipGetDevice(&callerDevice);
    std::vector<std::thread> threads;
    try {
        for (size_t i = 1; i < workers; i++) threads.emplace_back(work, i);
    } catch (...) {
        for (auto &t : threads) t.join();
        return PFAC_STATUS_ALLOC_FAILED;
    }
    work(0);
    for (auto &t : threads) t.join();
    (void)hipSetDevice(callerDevice);
    for (PFAC_status_t st : status)
        if (st != PFAC_STATUS_SUCCESS) return st;
    return PFAC_STATUS_SUCCESS;
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
    std::lock_guard<std::mutex> guard(handle->lock);          /* the match counter and the sort scratch belong to the handle */
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
    std::lock_guard<std::mutex> guard(handle->lock);

    return matchHostReduceOnGpu(handle, h_inputString, size, h_matched_result, h_pos, h_num_matched);
}

/* ------------------------------------------------------------- extensions */

PFAC_status_t PFACX_getInfo(PFAC_handle_t handle, PFACX_info_t *info)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!info || info->structSize < sizeof(size_t) || info->structSize > (size_t(1) << 16)) return PFAC_STATUS_INVALID_PARAMETER;
    const size_t callerSize = info->structSize;              /* a caller built against an older header passes a shorter struct: never written past */
    PFACX_info_t v;
    std::memset(&v, 0, sizeof(v));
    {
        std::lock_guard<std::mutex> guard(handle->lock);      /* setters swap the vectors read here */
        v.numOfPatterns = handle->fa.numPatterns;
        v.numOfStates = handle->fa.numStates;
        v.numOfFinalStates = handle->fa.numPatterns;
        v.initialState = handle->fa.initialState;
        v.maxPatternLen = handle->fa.maxPatternLen;
        v.numOfLeaves = handle->fa.numLeaves;
        v.perfMode = handle->perfMode;
        v.textureMode = handle->textureMode;
        v.platform = handle->platform;
        v.hasDevice = handle->hasDevice ? 1 : 0;
        v.numOfTableEntry = handle->numOfTableEntry;
        v.sizeOfTableEntry = handle->sizeOfTableEntry;
        v.sizeOfTableInBytes = handle->sizeOfTableInBytes;
        v.filterLog2Bits = handle->filter.log2Bits;
        v.filterHasShort = handle->filter.hasShort ? 1 : 0;
        v.filterBitsSet = handle->filter.bitsSet;
        v.kernelVariant = handle->kernelVariant;
        v.filterLog2BitsLadder = handle->filter.log2BitsLad;
        v.filterLog2BitsFinal3 = handle->filter.log2BitsF3;
        v.filterBitsSetLadder = handle->filter.bitsSetLad;
        v.ladderStops = handle->filter.ladderStops;
        v.ladderGoOns = handle->filter.ladderGoOns;
        v.ladderThin = handle->filter.ladderThin;
        v.ladderExtend = handle->filter.ladderExtend;
        v.trailingBytesIgnored = handle->fa.trailingBytes;
        v.chainJumpLog2 = handle->h_chainSlots.empty() ? 0 : handle->chainJumpLog2;
        v.chainSlots = handle->h_chainSlots.size();
        v.multiProcessorCount = handle->multiProcessorCount;
        /* what the pattern set holds on the device: the chained table, the initial row, the prefilter bitmaps, the launch
         * counters -- and the reference-layout table only while PFACX_KERNEL_REFTABLE has asked for it */
        size_t dev = 0;
        if (handle->d_chainSlots) dev += handle->numChainSlots * sizeof(pfac::ChainSlot);
        if (handle->d_dense) dev += handle->h_dense.size() * sizeof(int);
        if (handle->d_hashRow) dev += handle->h_hashRow.size() * sizeof(Int2);
        if (handle->d_hashVal) dev += handle->h_hashVal.size() * sizeof(Int2);
        if (handle->d_initialRow) dev += handle->h_initialRow.size() * sizeof(int);
        if (handle->d_gram3) dev += handle->filter.gram3.size() * sizeof(uint32_t);
        if (handle->d_ladder) dev += handle->filter.ladder.size() * sizeof(uint32_t);
        if (handle->d_final3) dev += handle->filter.final3.size() * sizeof(uint32_t);
        if (handle->d_shortBits) dev += handle->filter.shortBits.size() * sizeof(uint32_t);
        if (handle->d_gram1) dev += handle->filter.gram1.size() * sizeof(uint32_t);
        if (handle->d_prefix4) dev += handle->filter.prefix4.size() * sizeof(uint32_t);
        if (handle->d_workCounters) dev += pfac::kWorkCounterWords * sizeof(unsigned int);
        v.deviceTableBytes = dev;
        /* ... and what its calls have left allocated (grow-only, PFACX_trim gives it back): the two staging pieces of the host
         * paths (input + ids + positions: 9 bytes per position), the scratch the compacted output is ordered through, the list
         * of pattern-dense chunks */
        size_t scratch = 0;
        if (handle->hostStagePositions) scratch += 2 * (((handle->hostStagePositions + 3) & ~size_t(3)) + 2 * handle->hostStagePositions * sizeof(int));
        scratch += handle->reduceScratchBytes + handle->denseListEntries * sizeof(unsigned int);
        v.deviceScratchBytes = scratch;
        if (handle->h_modeHint) {
            v.streamNearMisses = (int)static_cast<volatile const unsigned int *>(handle->h_modeHint)[0];
            v.streamDense = (int)static_cast<volatile const unsigned int *>(handle->h_modeHint)[1];
        }
    }
    v.structSize = callerSize < sizeof(v) ? callerSize : sizeof(v);
    std::memcpy(info, &v, v.structSize);
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFACX_getTable(PFAC_handle_t handle, PFACX_table_t which, const void **ptr, size_t *bytes)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!ptr || !bytes) return PFAC_STATUS_INVALID_PARAMETER;
    std::lock_guard<std::mutex> guard(handle->lock);          /* setters free and rebuild what is handed out here */
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    *ptr = nullptr; *bytes = 0;
    switch (which) {
    case PFACX_TABLE_DENSE: {
        const PFAC_status_t st = ensureHostRefTable(handle);      /* TIME_DRIVEN: materialised on first use */
        if (st != PFAC_STATUS_SUCCESS) return st;
        *ptr = handle->h_dense.data(); *bytes = handle->h_dense.size() * sizeof(int); break;
    }
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
    case PFACX_TABLE_FILTER_LADDER:
        *ptr = handle->filter.ladder.data(); *bytes = handle->filter.ladder.size() * sizeof(uint32_t); break;
    case PFACX_TABLE_FILTER_FINAL3:
        *ptr = handle->filter.final3.data(); *bytes = handle->filter.final3.size() * sizeof(uint32_t); break;
    case PFACX_TABLE_FILTER_GRAM1:
        *ptr = handle->filter.gram1.data(); *bytes = handle->filter.gram1.size() * sizeof(uint32_t); break;
    case PFACX_TABLE_FILTER_PREFIX4:
        *ptr = handle->filter.prefix4.data(); *bytes = handle->filter.prefix4.size() * sizeof(uint32_t); break;
    case PFACX_TABLE_CHAIN: {
        if (handle->h_chainSlots.empty()) {
            const PFAC_status_t st = uploadChainedHashTable(handle);      /* host-only handle: builds, uploads nothing */
            if (st != PFAC_STATUS_SUCCESS) return st;
        }
        *ptr = handle->h_chainSlots.data(); *bytes = handle->h_chainSlots.size() * sizeof(pfac::ChainSlot); break;
    }
    default: return PFAC_STATUS_INVALID_PARAMETER;
    }
    return PFAC_STATUS_SUCCESS;
}

} /* extern "C" */

/* ---- compiled pattern sets on disk: the reference rebuilds everything per run
 * (PFAC_reorder_Table.cpp:121-231, PFAC.cpp:653-735).  File = header + tagged sections; the dense table is not
 * stored (S KiB: 0.5 GB for the 30 k-pattern set), it is refilled from the edges in a fraction of a second. */
namespace {

constexpr char kCompiledMagic[8] = {'P', 'F', 'A', 'C', 'X', 'C', '1', 0};
constexpr uint32_t kCompiledVersion = 7;          /* 2: patterns of 1-2 bytes are folded into the 3-gram bitmap; 3: root bucket + jump table behind the chained slots;
                                                     5: the prefix ladder replaces the 4-gram bitmap, the chained table has its own compact breadth-first layout;
                                                     6: 36-byte walk-queue entries (layout fingerprint); 7: no transition table is stored any more -- hashed and
                                                     chained tables are rebuilt from the checked trie at load (a file cannot steer a device read) --, the scalars
                                                     carry the pattern file's ignored trailing bytes */
/* what the stored tables depend on besides the patterns: hash constants and slot layout */
constexpr uint32_t kLayoutFingerprint = pfac::kGram3Mul ^ (pfac::kLadMul0 * 3u) ^ (pfac::kLadMul * 5u) ^ (pfac::kLadMulS * 11u) ^ (pfac::kLadMulG * 13u) ^ (pfac::kLadMulG2 * 17u) ^ (pfac::kFinal3Mul * 7u) ^ (pfac::kFinal3Mul2 * 19u) ^
                                        ((uint32_t)pfac::kLadderLevels << 12) ^
                                        ((uint32_t)sizeof(pfac::ChainSlot) << 24) ^ ((uint32_t)pfac::kChainMax << 20) ^ 0x20u /* entry bytes */ ^
                                        0x4000u /* chained table: multiply-shift bucket hash */;
struct CompiledHeader {
    char magic[8];
    uint32_t version, fingerprint, perfMode, jumpLog2;   /* jumpLog2: log2 of the jump-table slots at the end of the chained table */
    uint64_t payloadBytes, payloadFnv1a;
};
enum Section : uint32_t { kSecFile = 1, kSecScalars, kSecPatOff, kSecPatLen, kSecSorted, kSecEdgeBegin, kSecEdgeCh, kSecEdgeNext,
                          kSecFilter, kSecGram3, kSecLadder, kSecFinal3, kSecShort, kSecHashRow, kSecHashVal, kSecChain, kSecRootUnused, kSecInitialRow };

uint64_t fnv1a64(const unsigned char *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

template <class T>
void putSection(std::vector<unsigned char> &out, uint32_t tag, const T *data, size_t count)
{
    const uint64_t bytes = (uint64_t)count * sizeof(T);
    const unsigned char *t = reinterpret_cast<const unsigned char *>(&tag), *b = reinterpret_cast<const unsigned char *>(&bytes);
    out.insert(out.end(), t, t + 4);
    out.insert(out.end(), b, b + 8);
    const unsigned char *d = reinterpret_cast<const unsigned char *>(data);
    out.insert(out.end(), d, d + bytes);
}

template <class T>
bool takeSection(const unsigned char *p, uint64_t bytes, std::vector<T> &v)
{
    if (bytes % sizeof(T)) return false;
    v.resize(bytes / sizeof(T));
    if (bytes) std::memcpy(v.data(), p, bytes);
    return true;
}

} // namespace

extern "C" {

PFAC_status_t PFACX_saveCompiled(PFAC_handle_t handle, const char *filename)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!filename) return PFAC_STATUS_INVALID_PARAMETER;
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    std::lock_guard<std::mutex> guard(handle->lock);
    PFAC_context *c = handle;
    try {
        const pfac::Automaton &fa = c->fa;
        const pfac::Filter &f = c->filter;
        std::vector<unsigned char> payload;
        putSection(payload, kSecFile, fa.file.data(), fa.file.size());
        const int64_t scalars[6] = {fa.numPatterns, fa.maxPatternLen, fa.initialState, fa.numStates, fa.numLeaves, (int64_t)fa.trailingBytes};
        putSection(payload, kSecScalars, scalars, 6);
        putSection(payload, kSecPatOff, fa.patternOff.data(), fa.patternOff.size());
        putSection(payload, kSecPatLen, fa.patternLen.data(), fa.patternLen.size());
        putSection(payload, kSecSorted, fa.sortedId.data(), fa.sortedId.size());
        putSection(payload, kSecEdgeBegin, fa.edgeBegin.data(), fa.edgeBegin.size());
        putSection(payload, kSecEdgeCh, fa.edgeCh.data(), fa.edgeCh.size());
        putSection(payload, kSecEdgeNext, fa.edgeNext.data(), fa.edgeNext.size());
        const uint64_t filt[10] = {(uint64_t)f.log2Bits, (uint64_t)f.log2BitsLad, (uint64_t)f.log2BitsF3, f.hasShort ? 1u : 0u, f.bitsSet, f.bitsSetLad,
                                  f.ladderStops, f.ladderGoOns, (uint64_t)f.ladderThin, (uint64_t)f.ladderExtend};
        putSection(payload, kSecFilter, filt, 10);
        putSection(payload, kSecGram3, f.gram3.data(), f.gram3.size());
        putSection(payload, kSecLadder, f.ladder.data(), f.ladder.size());
        putSection(payload, kSecFinal3, f.final3.data(), f.final3.size());
        putSection(payload, kSecShort, f.shortBits.data(), f.shortBits.size());
        /* no transition table: dense, hashed and chained tables and the initial row are rebuilt from the edges at load */
        CompiledHeader h;
        std::memset(&h, 0, sizeof(h));
        std::memcpy(h.magic, kCompiledMagic, 8);
        h.version = kCompiledVersion; h.fingerprint = kLayoutFingerprint; h.perfMode = (uint32_t)c->perfMode;
        h.jumpLog2 = 0;                                        /* (was: log2 of the stored chained table's jump slots) */
        h.payloadBytes = payload.size(); h.payloadFnv1a = fnv1a64(payload.data(), payload.size());
        FILE *fp = std::fopen(filename, "wb");
        if (!fp) return PFAC_STATUS_FILE_OPEN_ERROR;
        const bool ok = std::fwrite(&h, sizeof(h), 1, fp) == 1 && (payload.empty() || std::fwrite(payload.data(), payload.size(), 1, fp) == 1);
        return (std::fclose(fp) == 0 && ok) ? PFAC_STATUS_SUCCESS : PFAC_STATUS_INTERNAL_ERROR;
    } catch (const std::bad_alloc &) { return PFAC_STATUS_ALLOC_FAILED; }
}

PFAC_status_t PFACX_loadCompiled(PFAC_handle_t handle, const char *filename)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!filename) return PFAC_STATUS_INVALID_PARAMETER;
    FILE *fp = std::fopen(filename, "rb");
    if (!fp) return PFAC_STATUS_FILE_OPEN_ERROR;
    CompiledHeader h;
    std::vector<unsigned char> payload;
    bool ok = std::fread(&h, sizeof(h), 1, fp) == 1 && std::memcmp(h.magic, kCompiledMagic, 8) == 0 && h.version == kCompiledVersion &&
              h.fingerprint == kLayoutFingerprint && (h.perfMode == PFAC_TIME_DRIVEN || h.perfMode == PFAC_SPACE_DRIVEN) &&
              h.payloadBytes < (uint64_t(1) << 40);
    try {
        if (ok) {
            payload.resize((size_t)h.payloadBytes);
            ok = payload.empty() || std::fread(payload.data(), payload.size(), 1, fp) == 1;
        }
    } catch (const std::bad_alloc &) { std::fclose(fp); return PFAC_STATUS_ALLOC_FAILED; }
    std::fclose(fp);
    if (!ok || fnv1a64(payload.data(), payload.size()) != h.payloadFnv1a) return PFAC_STATUS_INVALID_PARAMETER;   /* not a compiled set of this build, or damaged */

    /* everything is parsed and checked in temporaries: a refused file leaves the handle as it was.  The checksum is no
     * protection against a crafted file (FNV-1a is recomputed in a line), so nothing that a kernel indexes memory with is
     * taken from the file: the trie is checked to BE a trie of the stored patterns' depth, and every transition table --
     * hashed, chained, the initial row -- is rebuilt from it.  The prefilter bitmaps are taken as they are: they are read
     * with masked LDS addresses, a wrong bit can cost a result, not a memory access. */
    pfac::Automaton fa;
    pfac::Filter f;
    std::vector<int64_t> scalars;
    std::vector<uint64_t> filt;
    try {
        size_t at = 0;
        while (ok && at + 12 <= payload.size()) {
            uint32_t tag; uint64_t bytes;
            std::memcpy(&tag, &payload[at], 4); std::memcpy(&bytes, &payload[at + 4], 8);
            at += 12;
            if (bytes > payload.size() - at) { ok = false; break; }
            const unsigned char *p = payload.data() + at;
            switch (tag) {
            case kSecFile: ok = takeSection(p, bytes, fa.file); break;
            case kSecScalars: ok = takeSection(p, bytes, scalars); break;
            case kSecPatOff: ok = takeSection(p, bytes, fa.patternOff); break;
            case kSecPatLen: ok = takeSection(p, bytes, fa.patternLen); break;
            case kSecSorted: ok = takeSection(p, bytes, fa.sortedId); break;
            case kSecEdgeBegin: ok = takeSection(p, bytes, fa.edgeBegin); break;
            case kSecEdgeCh: ok = takeSection(p, bytes, fa.edgeCh); break;
            case kSecEdgeNext: ok = takeSection(p, bytes, fa.edgeNext); break;
            case kSecFilter: ok = takeSection(p, bytes, filt); break;
            case kSecGram3: ok = takeSection(p, bytes, f.gram3); break;
            case kSecLadder: ok = takeSection(p, bytes, f.ladder); break;
            case kSecFinal3: ok = takeSection(p, bytes, f.final3); break;
            case kSecShort: ok = takeSection(p, bytes, f.shortBits); break;
            default: break;                                    /* unknown section of a later writer: skipped */
            }
            at += (size_t)bytes;
        }
        ok = ok && scalars.size() == 6 && filt.size() == 10;
        for (size_t i = 0; ok && i < scalars.size(); i++) ok = scalars[i] >= 0 && scalars[i] < (int64_t(1) << 31);
        if (ok) {
            fa.numPatterns = (int)scalars[0]; fa.maxPatternLen = (int)scalars[1]; fa.initialState = (int)scalars[2];
            fa.numStates = (int)scalars[3]; fa.numLeaves = (int)scalars[4]; fa.trailingBytes = (size_t)scalars[5];
            f.log2Bits = (int)filt[0]; f.log2BitsLad = (int)filt[1]; f.log2BitsF3 = (int)filt[2]; f.hasShort = filt[3] != 0;
            f.bitsSet = (size_t)filt[4]; f.bitsSetLad = (size_t)filt[5];
            f.ladderStops = (size_t)filt[6]; f.ladderGoOns = (size_t)filt[7]; f.ladderThin = (int)filt[8]; f.ladderExtend = (int)filt[9];
            const size_t S = (size_t)fa.numStates, F = (size_t)fa.numPatterns;
            ok = fa.numStates > 0 && fa.initialState == fa.numPatterns + 1 && (size_t)fa.initialState < S &&
                 fa.patternOff.size() == F + 1 && fa.patternLen.size() == F + 1 && fa.sortedId.size() == F &&
                 fa.edgeBegin.size() == S + 1 && fa.edgeCh.size() == fa.edgeNext.size() && !fa.edgeBegin.empty() &&
                 fa.edgeBegin[0] == 0 && (size_t)fa.edgeBegin.back() == fa.edgeCh.size() && fa.trailingBytes <= fa.file.size() &&
                 f.log2Bits >= 13 && f.log2Bits <= 18 && f.log2BitsLad >= 13 && f.log2BitsLad <= 19 && f.log2BitsF3 >= 10 && f.log2BitsF3 <= 13 &&
                 pfac::kGram3LdsBytes + ((size_t(1) << f.log2BitsLad) + (size_t(1) << f.log2BitsF3)) / 8 + (f.hasShort ? 8192u : 0u) <= pfac::kFilterLdsBudget &&
                 f.gram3.size() == (size_t(1) << f.log2Bits) / 32 && f.ladder.size() == (size_t(1) << f.log2BitsLad) / 32 &&
                 f.final3.size() == (size_t(1) << f.log2BitsF3) / 32 && f.shortBits.size() == 65536 / 32;
            for (size_t i = 0; ok && i + 1 < fa.edgeBegin.size(); i++) ok = fa.edgeBegin[i] <= fa.edgeBegin[i + 1] && fa.edgeBegin[i] >= 0 && fa.edgeBegin[i + 1] - fa.edgeBegin[i] <= pfac::kCharSet;
            for (size_t i = 0; ok && i < fa.edgeNext.size(); i++) ok = fa.edgeNext[i] > 0 && (size_t)fa.edgeNext[i] < S && fa.edgeNext[i] != fa.initialState;
            /* what the kernels and the host path take on trust: the longest pattern (overlap of pieces and slices, the
             * safety margin at the end of the input), the pattern lengths and offsets */
            int longest = 0;
            for (size_t id = 1; ok && id <= F; id++) {
                ok = fa.patternLen[id] >= 1 && fa.patternOff[id] >= 0 && (size_t)fa.patternOff[id] + (size_t)fa.patternLen[id] <= fa.file.size();
                longest = fa.patternLen[id] > longest ? fa.patternLen[id] : longest;
            }
            ok = ok && fa.maxPatternLen == longest;
            /* the edges form a TREE below the initial state: every state is entered by at most one edge, the bytes of a
             * state's edges are distinct, no state lies deeper than the longest pattern (so no walk is longer: a cycle would
             * take a walk past the margin the kernels keep at the end of the input), and final state `id` lies exactly
             * patternLen[id] deep */
            if (ok) {
                std::vector<int> depth(S, -1);
                std::vector<int> order;
                order.reserve(S);
                depth[(size_t)fa.initialState] = 0;
                order.push_back(fa.initialState);
                for (size_t at2 = 0; ok && at2 < order.size(); at2++) {
                    const int st = order[at2];
                    uint64_t seen[4] = {0, 0, 0, 0};
                    for (int e = fa.edgeBegin[(size_t)st]; ok && e < fa.edgeBegin[(size_t)st + 1]; e++) {
                        const unsigned ch = fa.edgeCh[(size_t)e];
                        const int nx = fa.edgeNext[(size_t)e];
                        ok = !(seen[ch >> 6] & (uint64_t(1) << (ch & 63))) && depth[(size_t)nx] < 0 && depth[(size_t)st] < fa.maxPatternLen;
                        seen[ch >> 6] |= uint64_t(1) << (ch & 63);
                        if (ok) { depth[(size_t)nx] = depth[(size_t)st] + 1; order.push_back(nx); }
                    }
                }
                for (size_t id = 1; ok && id <= F; id++) ok = depth[id] == fa.patternLen[id];
                /* states the initial state does not reach must have no edges (state 0 is the unused one) */
                for (size_t st = 0; ok && st < S; st++) ok = depth[st] >= 0 || fa.edgeBegin[st] == fa.edgeBegin[st + 1];
            }
        }
    } catch (const std::bad_alloc &) { return PFAC_STATUS_ALLOC_FAILED; }
    if (!ok) return PFAC_STATUS_INVALID_PARAMETER;

    std::lock_guard<std::mutex> guard(handle->lock);
    std::unique_lock<std::shared_mutex> tables(handle->tablesInUse);
    PFAC_context *c = handle;
    if (c->isPatternsReady) freeResources(c);
    c->patternFile = filename;
    c->perfMode = (int)h.perfMode;
    c->fa = std::move(fa);
    /* The prefilter bitmaps are rebuilt from the checked trie as well, like every table (30 ms for a Snort-scale set): a stale or
     * crafted file with a valid checksum could not make a kernel read outside a bitmap (addresses are masked), but a cleared
     * bit silently drops matches, and the full-result path (gram3 / ladder from the file) could disagree with the compacted-
     * output path (gram1 / prefix4, always rebuilt).  The file's copies are read, size-checked and dropped. */
    c->filter = pfac::Filter();
    c->isPatternsReady = true;
    pfac::buildInitialRow(c->fa, c->h_initialRow);
    PFAC_status_t st;
    try { st = bindCommon(c, /*build=*/true); } catch (const std::bad_alloc &) { st = PFAC_STATUS_ALLOC_FAILED; }
    if (st == PFAC_STATUS_SUCCESS) st = bindTable(c);
    if (st != PFAC_STATUS_SUCCESS) { freeResources(c); return st; }
    return PFAC_STATUS_SUCCESS;
}

/* pfac_ext.h: give back the grow-only device buffers of the handle (staging of PFAC_matchFromHost, copies of
 * PFAC_matchFromHostReduce, sort scratch, the dense-chunk list); the next call that needs one allocates it again */
PFAC_status_t PFACX_trim(PFAC_handle_t handle)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    std::lock_guard<std::mutex> guard(handle->lock);
    freeHostStage(handle);
    devFree(handle->d_reduceScratch);
    handle->reduceScratchBytes = 0;
    devFree(handle->d_denseList);
    handle->denseListEntries = 0;
    for (auto &child : handle->children) if (child.second) (void)PFACX_trim(child.second);
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFACX_getScanStats(PFAC_handle_t handle, PFACX_scan_stats_t *stats)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!stats || stats->structSize < sizeof(size_t) || stats->structSize > (size_t(1) << 16)) return PFAC_STATUS_INVALID_PARAMETER;
    const size_t callerSize = stats->structSize < sizeof(PFACX_scan_stats_t) ? stats->structSize : sizeof(PFACX_scan_stats_t);
    PFACX_scan_stats_t local;
    struct CopyOut {                                          /* whatever the outcome: the caller's struct, as far as it reaches */
        PFACX_scan_stats_t *dst, *src; size_t n;
        ~CopyOut() { src->structSize = n; std::memcpy(dst, src, n); }
    } copyOut{stats, &local, callerSize};
    stats = &local;
    std::memset(stats, 0, sizeof(*stats));
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    std::lock_guard<std::mutex> guard(handle->lock);
    if (!handle->hasDevice || !handle->d_workCounters) return PFAC_STATUS_LIB_NOT_EXIST;
    unsigned long long v[pfac::kStatsCount + 3];             /* published by the last block of the launch: scan_gfx950.hip, the kernel's end */
    if (hipStreamSynchronize(nullptr) != hipSuccess ||
        hipMemcpy(v, handle->d_workCounters + pfac::kStatsPublishedWord, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
        return PFAC_STATUS_INTERNAL_ERROR;
    stats->walkerRounds = v[0]; stats->laneSteps = v[1]; stats->walksStarted = v[2]; stats->level1Hits = v[3];
    stats->ladderCandidates = v[5];
    const unsigned long long dense = v[pfac::kStatsCount];
    stats->denseChunks = dense;
    stats->filterKernelMs = -1.0;
    if (handle->kernelTiming && handle->evTimeRecorded && handle->evTime[0] && handle->evTime[1]) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, static_cast<hipEvent_t>(handle->evTime[0]), static_cast<hipEvent_t>(handle->evTime[1])) == hipSuccess) stats->filterKernelMs = ms;
        else (void)hipGetLastError();
    }
    stats->tilesPerChunk = pfac::kChunkTiles;
    stats->walksPerLane = (int)v[pfac::kStatsCount + 1];     /* of the launch the counters describe: the full-result and the compacted-output kernel differ */
    stats->stageModeWaves = v[pfac::kStatsCount + 2] & 0xFFFFFFFFull;
    stats->walker = (v[pfac::kStatsCount + 2] >> 32) ? PFACX_WALKER_STAGE : PFACX_WALKER_WINDOW;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFACX_setKernelTiming(PFAC_handle_t handle, int on)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    std::lock_guard<std::mutex> guard(handle->lock);
    if (!handle->hasDevice) return PFAC_STATUS_LIB_NOT_EXIST;
    handle->evTimeRecorded = false;
    if (on) {
        for (auto &e : handle->evTime)
            if (!e) {
                hipEvent_t ev = nullptr;
                if (hipEventCreate(&ev) != hipSuccess) { (void)hipGetLastError(); return PFAC_STATUS_INTERNAL_ERROR; }
                e = ev;
            }
    }
    handle->kernelTiming = on != 0;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFACX_setWalker(PFAC_handle_t handle, int walker)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (walker != PFACX_WALKER_AUTO && walker != PFACX_WALKER_WINDOW && walker != PFACX_WALKER_STAGE) return PFAC_STATUS_INVALID_PARAMETER;
    std::lock_guard<std::mutex> guard(handle->lock);
    handle->walker = walker;
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFACX_setKernelVariant(PFAC_handle_t handle, int variant)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (variant != PFACX_KERNEL_FILTER && variant != PFACX_KERNEL_NAIVE && variant != PFACX_KERNEL_AUTO && variant != PFACX_KERNEL_REFTABLE)
        return PFAC_STATUS_INVALID_PARAMETER;
    std::lock_guard<std::mutex> guard(handle->lock);
    if (variant == PFACX_KERNEL_REFTABLE && handle->isPatternsReady) {      /* the one kernel that reads the reference-layout table on the device */
        const PFAC_status_t st = ensureDeviceRefTable(handle);
        if (st != PFAC_STATUS_SUCCESS) return st;
    }
    handle->kernelVariant = variant;
    return PFAC_STATUS_SUCCESS;
}

} /* extern "C" */

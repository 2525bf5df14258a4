/*
 * pfac_api.cpp -- the reference-compatible C ABI of libpfac.so (include/PFAC.h)
 * plus the PFACX_ extensions (include/pfac_ext.h).
 *
 * Mirrors the observable behaviour of PFAC/src/PFAC.cpp: argument-check order,
 * status codes, lifecycle (a second readPatternFromFile replaces the first,
 * setPerfMode after load rebuilds the table), the dlopen'd kernel-module seam
 * and the per-call device temporaries of PFAC_matchFromHost.  HIP replaces the
 * CUDA runtime 1:1 on the host side; all device work is in the module
 * (scan_*.hip).
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

namespace pfac_internal {


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
    devFree(c->d_denseFast);
    c->denseFastEntries = 0;
    devFree(c->d_chainNarrow);
    c->numChainNarrow = 0;
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
    devFree(c->d_tail);
    devFree(c->d_workCounters);
    if (c->h_modeHint) { (void)hipHostFree(c->h_modeHint); c->h_modeHint = c->d_modeHint = nullptr; }
    devFree(c->d_reduceScratch);
    c->orderCleanBase = nullptr;
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
    if (st != PFAC_STATUS_SUCCESS) return st;
    {   /* the narrow form for the tiled kernel on text (pfac_context.h); without it that kernel walks the wide one */
        std::vector<pfac::ChainSlot> narrow;
        int lg = 0;
        if (pfac::buildChainedHashTable(c->fa, narrow, lg, /*narrow=*/true) == PFAC_STATUS_SUCCESS && !narrow.empty() &&
            upload(c->d_chainNarrow, narrow.data(), narrow.size()) == PFAC_STATUS_SUCCESS) {
            c->numChainNarrow = narrow.size();
            c->chainNarrowJumpLog2 = lg;
        } else {
            devFree(c->d_chainNarrow);
            c->numChainNarrow = 0;
            (void)hipGetLastError();
        }
    }
    /* the dense table next to it for a small set that does not fold (PFAC_context::d_denseFast): states inside chains -- one way on, nothing ends there --
     * are what the chained table saves steps on; a set with less than a quarter of them keeps int[S][256] too */
    const pfac::Automaton &fa = c->fa;
    if (fa.numStates > fa.initialState && fa.numStates <= pfac::kDenseFastMaxStates) {
        size_t inside = 0;
        for (int s = fa.initialState + 1; s < fa.numStates; s++) inside += fa.edgeBegin[s + 1] - fa.edgeBegin[s] == 1 ? 1u : 0u;      /* (final states are numbered below the initial state) */
        if (inside * 4 < (size_t)fa.numStates) {
            std::vector<int> dense;
            if (!c->h_dense.empty()) dense = c->h_dense;
            else if (pfac::buildDenseTable(fa, dense) != PFAC_STATUS_SUCCESS) dense.clear();
            if (!dense.empty() && upload(c->d_denseFast, dense.data(), dense.size()) == PFAC_STATUS_SUCCESS) c->denseFastEntries = dense.size();
            else { devFree(c->d_denseFast); c->denseFastEntries = 0; (void)hipGetLastError(); }      /* no table: AUTO keeps to the chained one */
        }
    }
    return PFAC_STATUS_SUCCESS;
}

/* The reference-layout table of the perf mode on the HOST: the dense table is materialised on first use -- PFACX_getTable,
 * the CPU platforms, PFACX_KERNEL_REFTABLE -- because neither GPU kernel of the product path reads it (both walk the
 * chained table) and it is S KiB: 498 MB for a Snort-scale set.  The hashed tables (a few MB) are built with the set. */
static PFAC_status_t ensureHostRefTable(PFAC_context *c, bool tablesLocked = false)
{
    if (c->perfMode == PFAC_TIME_DRIVEN && c->h_dense.empty()) {
        if (tablesLocked) return pfac::buildDenseTable(c->fa, c->h_dense);
        std::unique_lock<std::shared_mutex> w(c->tablesInUse);         /* (the caller holds c->lock) */
        return pfac::buildDenseTable(c->fa, c->h_dense);
    }
    return PFAC_STATUS_SUCCESS;
}

/* ... and on the DEVICE: only the reference-shaped kernel (PFACX_KERNEL_REFTABLE) reads it there */
static PFAC_status_t ensureDeviceRefTable(PFAC_context *c, bool tablesLocked = false)
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
PFAC_status_t bindCommon(PFAC_context *c, bool build)
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
    if (st == PFAC_STATUS_SUCCESS && !c->filter.tail.empty()) st = upload(c->d_tail, c->filter.tail.data(), c->filter.tail.size());
    else if (st == PFAC_STATUS_SUCCESS && !c->filter.tailG.empty()) st = upload(c->d_tail, c->filter.tailG.data(), c->filter.tailG.size());
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
            /* every word: [1] routes PFACX_KERNEL_AUTO, [4] / [5] are the pair count and the sequence number a compacted-output call polls
             * (sequence numbers are never 0); hipHostMalloc does not promise zeros, and a block given back by freeResources may come back stale */
            std::memset(h, 0, 64);
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

/* ... for a caller that holds c->lock and drives several threads through one handle (multi_gpu.cpp: the workers of a CPU-platform handle):
 * the tables first, once; then any number of threads may match side by side */
PFAC_status_t prepareCpuPlatformLocked(PFAC_context *c) { return ensureHostRefTable(c); }
PFAC_status_t matchHostOnCpuPlatformPrepared(PFAC_context *c, const char *in, size_t n, int *out)
{
    bool omp = false;
    if (c->platform == PFAC_PLATFORM_CPU_OMP) omp = (std::getenv("OMP_NUM_THREADS") != nullptr);
    std::shared_lock<std::shared_mutex> r(c->tablesInUse);
    if (c->perfMode == PFAC_TIME_DRIVEN && c->h_dense.empty()) return PFAC_STATUS_PATTERNS_NOT_READY;
    return pfac::matchOnCpu(c, reinterpret_cast<const unsigned char *>(in), n, out, omp);
}

} // namespace pfac_internal
using namespace pfac_internal;

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

    return matchHostReduceOnGpu(handle, h_inputString, size, size, 0, h_matched_result, h_pos, h_num_matched);
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
        v.filterLadderLast = handle->filter.ladderLast;
        v.filterTailEntries = handle->filter.tailEntries;
        v.filterTailGlobalEntries = handle->filter.tailGEntries;
        v.filterLog2TailGlobal = handle->filter.tailG.empty() ? 0 : handle->filter.log2TailG;
        v.filterSkipTags = handle->filter.skipCount;
        v.filterLadderSalt = handle->filter.ladderSalt;
        v.trailingBytesIgnored = handle->fa.trailingBytes;
        v.chainJumpLog2 = handle->h_chainSlots.empty() ? 0 : handle->chainJumpLog2;
        v.chainSlots = handle->h_chainSlots.size();
        v.multiProcessorCount = handle->multiProcessorCount;
        /* what the pattern set holds on the device: the chained table, the initial row, the prefilter bitmaps, the launch
         * counters -- and the reference-layout table only while PFACX_KERNEL_REFTABLE has asked for it */
        size_t dev = 0;
        if (handle->d_chainSlots) dev += handle->numChainSlots * sizeof(pfac::ChainSlot);
        if (handle->d_chainNarrow) dev += handle->numChainNarrow * sizeof(pfac::ChainSlot);
        if (handle->d_dense) dev += handle->h_dense.size() * sizeof(int);
        if (handle->d_denseFast) dev += handle->denseFastEntries * sizeof(int);
        if (handle->d_hashRow) dev += handle->h_hashRow.size() * sizeof(Int2);
        if (handle->d_hashVal) dev += handle->h_hashVal.size() * sizeof(Int2);
        if (handle->d_initialRow) dev += handle->h_initialRow.size() * sizeof(int);
        if (handle->d_gram3) dev += handle->filter.gram3.size() * sizeof(uint32_t);
        if (handle->d_ladder) dev += handle->filter.ladder.size() * sizeof(uint32_t);
        if (handle->d_final3) dev += handle->filter.final3.size() * sizeof(uint32_t);
        if (handle->d_shortBits) dev += handle->filter.shortBits.size() * sizeof(uint32_t);
        if (handle->d_gram1) dev += handle->filter.gram1.size() * sizeof(uint32_t);
        if (handle->d_prefix4) dev += handle->filter.prefix4.size() * sizeof(uint32_t);
        if (handle->d_tail) dev += (handle->filter.tail.size() + handle->filter.tailG.size()) * sizeof(uint32_t);
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
    case PFACX_TABLE_FILTER_TAIL:
        *ptr = handle->filter.tail.data(); *bytes = handle->filter.tail.size() * sizeof(uint32_t); break;
    case PFACX_TABLE_FILTER_TAIL_GLOBAL:
        *ptr = handle->filter.tailG.data(); *bytes = handle->filter.tailG.size() * sizeof(uint32_t); break;
    case PFACX_TABLE_FILTER_SKIP:
        *ptr = handle->filter.skipTags; *bytes = (size_t)handle->filter.skipCount * sizeof(uint32_t); break;
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
extern "C" {

/* pfac_ext.h: give back the grow-only device buffers of the handle (staging of PFAC_matchFromHost, copies of
 * PFAC_matchFromHostReduce, sort scratch, the dense-chunk list); the next call that needs one allocates it again */
PFAC_status_t PFACX_trim(PFAC_handle_t handle)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    std::lock_guard<std::mutex> guard(handle->lock);
    freeHostStage(handle);
    devFree(handle->d_reduceScratch);
    handle->reduceScratchBytes = 0;
    handle->orderCleanBase = nullptr;
    devFree(handle->d_denseList);
    handle->denseListEntries = 0;
    for (auto &child : handle->children) if (child.second) (void)PFACX_trim(child.second);
    return PFAC_STATUS_SUCCESS;
}

PFAC_status_t PFACX_prepare(PFAC_handle_t handle, size_t maxBytes)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    if (handle->platform != PFAC_PLATFORM_GPU) return PFAC_STATUS_SUCCESS;       /* the CPU platforms keep nothing between calls */
    std::lock_guard<std::mutex> guard(handle->lock);
    return prepareHostPath(handle, maxBytes);
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
    unsigned long long v[pfac::kStatsCount + 3];             /* published by the last block of the launch: scan_*.hip, the kernel's end */
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
    stats->walker = ((v[pfac::kStatsCount + 2] >> 32) & 1ull) ? PFACX_WALKER_STAGE : PFACX_WALKER_WINDOW;
    stats->veto = (int)((v[pfac::kStatsCount + 2] >> 33) & 3ull);
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
    if (walker != PFACX_WALKER_AUTO && walker != PFACX_WALKER_WINDOW && walker != PFACX_WALKER_STAGE && walker != PFACX_WALKER_VETO) return PFAC_STATUS_INVALID_PARAMETER;
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

/*
 * host_pipeline.cpp -- host buffers through the GPU: PFAC_matchFromHost and PFAC_matchFromHostReduce on the GPU platform
 * (ref PFAC/src/PFAC.cpp:879-961, 1010-1128: allocate, upload, scan, download, free, in sequence).  The stream goes through two
 * staging pieces owned by the handle: piece i + 1 uploads while piece i is scanned by the compacted-output kernel, only the
 * (position, id) pairs come back, the zeros of the result vector are written on the host.
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
#include <condition_variable>
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
using namespace pfac_internal;

namespace {

/* The threads of a host call wait for each other's progress -- the uploader for a scanned buffer, the caller for an upload to be queued and for
 * a piece of its vector to be filled -- on a condition variable: round 5 spun on std::this_thread::yield(), which on a host whose cores are all
 * busy (the zero fill runs up to eight threads beside the DMA engine's reads) takes the very cores the fill threads need.  Progress counters stay
 * atomics (the fast path is one acquire load); whoever advances one calls bump(). */
struct Progress {
    std::mutex m;
    std::condition_variable cv;
    void bump() { { std::lock_guard<std::mutex> g(m); } cv.notify_all(); }
    template <class Pred> void wait(Pred done)
    {
        if (done()) return;
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, done);
    }
};

} // namespace

namespace pfac_internal {

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

/* PFACX_prepare: everything a handle's first PFAC_matchFromHost / PFAC_matchFromHostReduce would otherwise allocate, create or load inside the
 * call (round 5's driver line: first call 103 ms, steady 7 ms): the two staging pieces with their streams and events, the ordering scratch of a
 * piece, the code objects of the compacted-output scan and its ordering launches (one throwaway scan of a piece filled with a byte no pattern
 * starts with), and the runtime's own staging of pageable host memory (one throwaway upload of a pageable piece).  The caller holds c->lock. */
PFAC_status_t prepareHostPath(PFAC_context *c, size_t maxBytes)
{
    if (!c->hasDevice || !c->module) return PFAC_STATUS_LIB_NOT_EXIST;
    const size_t overlap = (size_t)c->fa.maxPatternLen;
    const size_t want = maxBytes == 0 || maxBytes > kHostPiece ? kHostPiece : maxBytes;
    PFAC_status_t st = ensureHostStage(c, want + overlap);
    if (st != PFAC_STATUS_SUCCESS) return st;
    correctTextureMode(c);
    int filler = 0;                                            /* a byte the initial state has no transition on, if there is one: the scan then finds nothing */
    for (int b = 0; b < pfac::kCharSet && (size_t)b < c->h_initialRow.size(); b++)
        if (c->h_initialRow[(size_t)b] == pfac::kTrapState) { filler = b; break; }
    hipStream_t up = static_cast<hipStream_t>(c->stageUp);
    const size_t n = want + overlap;
    bool ok = true;
    try {
        const std::vector<char> pageable(n, (char)filler);
        ok = hipMemcpyAsync(c->d_stageIn[0], pageable.data(), n, hipMemcpyHostToDevice, up) == hipSuccess && hipStreamSynchronize(up) == hipSuccess &&
             hipMemsetAsync(c->d_stageIn[1], filler, n, nullptr) == hipSuccess;
    } catch (const std::bad_alloc &) { return PFAC_STATUS_ALLOC_FAILED; }
    if (!ok) { (void)hipGetLastError(); return PFAC_STATUS_INTERNAL_ERROR; }
    PFAC_reduce_kernel_protoType reduce = c->perfMode == PFAC_TIME_DRIVEN ? c->reduce_kernel_ptr : c->reduce_inplace_kernel_ptr;
    for (int b = 0; b < 2 && st == PFAC_STATUS_SUCCESS; b++) {              /* both buffers, the way both host calls use them: pairs in any order / in position order */
        int count = 0;
        c->reduceUnordered = b == 0;
        st = reduce(c, reinterpret_cast<int *>(c->d_stageIn[b]), (int)n, c->d_stageOut[b], c->d_stagePos[b], &count, nullptr, nullptr);
        c->reduceUnordered = false;
    }
    if (st == PFAC_STATUS_SUCCESS && hipStreamSynchronize(nullptr) != hipSuccess) st = PFAC_STATUS_INTERNAL_ERROR;
    return st;
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
    Progress progress;                                         /* what the threads of this call wait for each other on */
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
                if (hipSetDevice(device) != hipSuccess) { uploadFailed.store(true); progress.bump(); return; }
                for (size_t i = 0; i < numPieces; i++) {
                    if (i >= 2) progress.wait([&]() { return scansDone.load(std::memory_order_acquire) + 1 >= i || stopUploads.load(std::memory_order_relaxed); });
                    if (stopUploads.load(std::memory_order_relaxed)) return;
                    if (!uploadPiece(i)) { uploadFailed.store(true); progress.bump(); return; }
                    uploadsQueued.store(i + 1, std::memory_order_release);
                    progress.bump();
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
    /* read by the fill threads for as long as they run: declared where joinAll() still sees them */
    cpu_set_t fillCpus;
    CPU_ZERO(&fillCpus);
    bool bindFill = false;
    try {
        filled.reset(new std::atomic<unsigned>[numPieces]);
        for (size_t k = 0; k < numPieces; k++) filled[k].store(0, std::memory_order_relaxed);
        fillers.reserve(helpers);
        /* The fill threads run on the NUMA node the caller's result vector lives on: 4 bytes per position of streaming stores that
         * cross the sockets' link meet the link's own reads of the input there (2 x EPYC 9575F, GPU on node 0, pinned buffers
         * first-touched on node 1: p50 7.4 ms, p90 11.4 ms per 256 MiB call against 5.5 / 6.2 ms with the buffers on node 0 --
         * the driver's round-4 line: 29 GB/s median; tools/host_numa_probe.py).  PFAC_HOST_FILL_ANYWHERE=1 leaves them to the OS. */
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
                    progress.bump();
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
            progress.wait([&]() { return filled[k].load(std::memory_order_acquire) >= started; });
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
            progress.wait([&]() { return uploadsQueued.load(std::memory_order_acquire) > i || uploadFailed.load(std::memory_order_relaxed); });
            ok = !uploadFailed.load(std::memory_order_relaxed) && hipStreamWaitEvent(nullptr, static_cast<hipEvent_t>(c->evUp[b]), 0) == hipSuccess;
            if (!ok) break;
            int count = 0;
            c->reduceUnordered = true;
            st = reduce(c, reinterpret_cast<int *>(c->d_stageIn[b]), (int)scanned, c->d_stageOut[b], c->d_stagePos[b], &count, nullptr, nullptr);
            c->reduceUnordered = false;
            if (st != PFAC_STATUS_SUCCESS) break;
            scansDone.store(i + 1, std::memory_order_release);     /* the scan is synchronous: its input buffer may take piece i + 2 */
            progress.bump();
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
    progress.bump();
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
 * The stream may be a slice of a longer one (PFACX_matchFromHostReduceMultiGPU): positions [0, owned) get their pairs, `readable`
 * bytes may be read, posBase is added to every position; the caller's arrays hold `owned` entries.
 */
constexpr size_t kHostReducePiece = size_t(16) << 20;
PFAC_status_t matchHostReduceOnGpu(PFAC_context *c, char *h_inputString, size_t size, size_t readable, size_t posBase, int *h_matched_result, int *h_pos, int *h_num_matched)
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
        const size_t scanned = readable - off < mine + overlap ? readable - off : mine + overlap;
        return hipMemcpyAsync(c->d_stageIn[i & 1], h_inputString + off, scanned, hipMemcpyHostToDevice, up) == hipSuccess &&
               hipEventRecord(static_cast<hipEvent_t>(c->evUp[i & 1]), up) == hipSuccess;
    };
    std::atomic<size_t> scansDone{0}, uploadsQueued{0};
    std::atomic<bool> uploadFailed{false}, stopUploads{false};
    Progress progress;
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
                if (hipSetDevice(device) != hipSuccess) { uploadFailed.store(true); progress.bump(); return; }
                for (size_t i = 0; i < numPieces; i++) {
                    if (i >= 2) progress.wait([&]() { return scansDone.load(std::memory_order_acquire) + 1 >= i || stopUploads.load(std::memory_order_relaxed); });
                    if (stopUploads.load(std::memory_order_relaxed)) return;
                    if (!uploadPiece(i)) { uploadFailed.store(true); progress.bump(); return; }
                    uploadsQueued.store(i + 1, std::memory_order_release);
                    progress.bump();
                }
            });
        } catch (...) { ok = false; }
    }
    size_t total = 0;
    for (size_t i = 0; i < numPieces && ok && st == PFAC_STATUS_SUCCESS; i++) {
        const int b = (int)(i & 1);
        const size_t off = i * piece;
        const size_t mine = size - off < piece ? size - off : piece;
        const size_t scanned = readable - off < mine + overlap ? readable - off : mine + overlap;
        progress.wait([&]() { return uploadsQueued.load(std::memory_order_acquire) > i || uploadFailed.load(std::memory_order_relaxed); });
        ok = !uploadFailed.load(std::memory_order_relaxed) && hipStreamWaitEvent(nullptr, static_cast<hipEvent_t>(c->evUp[b]), 0) == hipSuccess;
        if (!ok) break;
        int count = 0;
        st = reduce(c, reinterpret_cast<int *>(c->d_stageIn[b]), (int)scanned, c->d_stageOut[b], c->d_stagePos[b], &count, nullptr, nullptr);
        if (st != PFAC_STATUS_SUCCESS) break;
        scansDone.store(i + 1, std::memory_order_release);     /* the scan is synchronous: its input buffer may take piece i + 2 */
        progress.bump();
        if (count == 0) continue;
        /* total <= off (a position has at most one pair); the pairs that stay (positions below `mine`) are at most `mine`, so they lie among the
         * first size - total of the list: the caller's arrays (size entries) hold what is copied */
        const size_t room = size - total, copied = (size_t)count < room ? (size_t)count : room;
        if (hipMemcpy(h_pos + total, c->d_stagePos[b], copied * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        size_t keep = copied;                                  /* positions ascend: those in the overlap are a suffix */
        while (keep > 0 && (size_t)h_pos[total + keep - 1] >= mine) keep--;
        if (keep && hipMemcpy(h_matched_result + total, c->d_stageOut[b], keep * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        if (off + posBase) for (size_t k = 0; k < keep; k++) h_pos[total + k] += (int)(off + posBase);
        total += keep;
    }
    if (!ok && st == PFAC_STATUS_SUCCESS) st = PFAC_STATUS_INTERNAL_ERROR;
    stopUploads.store(true);
    progress.bump();
    if (uploader.joinable()) uploader.join();
    const bool drained = hipStreamSynchronize(up) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess;
    if (!drained && st == PFAC_STATUS_SUCCESS) st = PFAC_STATUS_INTERNAL_ERROR;
    if (st == PFAC_STATUS_SUCCESS) *h_num_matched = (int)total;
    return st;
}

} // namespace pfac_internal

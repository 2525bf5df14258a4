/*
 * multi_gpu.cpp -- PFACX_matchFromHostMultiGPU (include/pfac_ext.h): one call shards a host stream over several GPUs of the node.
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

using namespace pfac_internal;

namespace {

/* The frame of both multi-GPU calls: one worker thread per listed device -- hipSetDevice, a per-device handle with this handle's pattern set and
 * modes (kept in the handle for the next call), a contiguous slice [bound[i], bound[i + 1]) of the stream (boundaries rounded to the 1 KiB tile:
 * pfac_amd/sharding.py plan_slices is the Python mirror) -- and `call(worker's handle, i, lo, hi)` on it, the worker's lock held. */
template <class Call>
PFAC_status_t onDevices(PFAC_handle_t handle, size_t size, int numDevices, const int *devices, std::vector<size_t> &bound, Call call)
{
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
    bound.assign(workers + 1, 0);
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
        w->walker = c->walker;                                     /* PFACX_setWalker on the parent reaches the workers */
        status[i] = call(w, i, bound[i], bound[i + 1]);
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

/* A handle on a CPU platform (PFAC_setPlatform, or PFACX_createHostOnly) has no devices to shard over; the two calls then run their `numDevices`
 * workers (0 = one) as host threads over the CPU matchers: the same slices, the same read-ahead, the same rebasing and moving-together of the
 * pairs -- what a machine without a GPU can exercise of this file (tests/test_host_api.py runs eight workers).  Worker i gets the results of
 * positions [bound[i], bound[i + 1]) in a vector of its own (its matcher reads, and writes results for, the maxPatternLen bytes behind its
 * slice too: the neighbour's positions). */
template <class Take>
PFAC_status_t onCpuWorkers(PFAC_handle_t handle, const char *in, size_t size, int numDevices, std::vector<size_t> &bound, Take take)
{
    std::lock_guard<std::mutex> guard(handle->lock);
    PFAC_context *c = handle;
    PFAC_status_t st = prepareCpuPlatformLocked(c);
    if (st != PFAC_STATUS_SUCCESS) return st;
    const size_t workers = numDevices > 0 ? (size_t)numDevices : 1;
    const size_t overlap = (size_t)c->fa.maxPatternLen;
    bound.assign(workers + 1, 0);
    for (size_t i = 1; i < workers; i++) {
        size_t b = (size * i / workers) / 1024 * 1024;
        bound[i] = b > bound[i - 1] ? b : bound[i - 1];
    }
    bound[workers] = size;
    std::vector<PFAC_status_t> status(workers, PFAC_STATUS_SUCCESS);
    auto work = [&](size_t i) {
        const size_t lo = bound[i], hi = bound[i + 1];
        if (hi == lo) return;
        const size_t scanned = size - lo < hi - lo + overlap ? size - lo : hi - lo + overlap;
        try {
            std::vector<int> res(scanned);
            status[i] = matchHostOnCpuPlatformPrepared(c, in + lo, scanned, res.data());
            if (status[i] == PFAC_STATUS_SUCCESS) take(i, lo, hi, res.data());
        } catch (const std::bad_alloc &) { status[i] = PFAC_STATUS_ALLOC_FAILED; }
    };
    std::vector<std::thread> threads;
    try {
        for (size_t i = 1; i < workers; i++) threads.emplace_back(work, i);
    } catch (...) {
        for (auto &t : threads) t.join();
        return PFAC_STATUS_ALLOC_FAILED;
    }
    work(0);
    for (auto &t : threads) t.join();
    for (PFAC_status_t s : status)
        if (s != PFAC_STATUS_SUCCESS) return s;
    return PFAC_STATUS_SUCCESS;
}

} // namespace

extern "C" {

/*
 * pfac_ext.h: one call shards a host stream over several GPUs (what every user of the
 * reference re-writes from PFAC/test/omp_PFAC.cpp:257-394 or SimpleMultiGPU_pthread.cpp:50-174).  Each worker's slice is
 * scanned together with the maxPatternLen bytes behind it, only the slice's own results written (omp_PFAC.cpp:324,377).
 * No exchange between devices.
 */
PFAC_status_t PFACX_matchFromHostMultiGPU(PFAC_handle_t handle, char *h_inputString, size_t size, int *h_matched_result,
                                          int numDevices, const int *devices)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    if (!h_inputString || !h_matched_result || numDevices < 0) return PFAC_STATUS_INVALID_PARAMETER;
    if (size == 0) return PFAC_STATUS_SUCCESS;
    std::vector<size_t> bound;
    if (handle->platform != PFAC_PLATFORM_GPU)
        return onCpuWorkers(handle, h_inputString, size, numDevices, bound, [&](size_t, size_t lo, size_t hi, const int *res) {
            std::memcpy(h_matched_result + lo, res, (hi - lo) * sizeof(int));
        });
    return onDevices(handle, size, numDevices, devices, bound, [&](PFAC_context *w, size_t, size_t lo, size_t hi) {
        return matchHostOnGpu(w, h_inputString + lo, hi - lo, size - lo, h_matched_result + lo);
    });
}

/*
 * pfac_ext.h: the compacted-output form -- what scales with the host links (1 B per position over a link, nothing filled on the
 * host: DESIGN.md 5).  Worker i leaves the pairs of its slice, positions counted from the start of the stream, at the slice's own
 * offset of the caller's arrays (it has at most one pair per position); the lists are then moved together: slices are in stream
 * order, so the whole list is in position order.
 */
PFAC_status_t PFACX_matchFromHostReduceMultiGPU(PFAC_handle_t handle, char *h_inputString, size_t size, int *h_matched_result, int *h_pos,
                                                int *h_num_matched, int numDevices, const int *devices)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    if (!h_inputString || !h_matched_result || !h_pos || !h_num_matched || numDevices < 0 || size > 0x7FFFFFFFull) return PFAC_STATUS_INVALID_PARAMETER;
    *h_num_matched = 0;
    if (size == 0) return PFAC_STATUS_SUCCESS;
    std::vector<size_t> bound;
    std::vector<int> counts;                                                     /* one per worker */
    PFAC_status_t st;
    if (handle->platform != PFAC_PLATFORM_GPU) {
        counts.assign((size_t)(numDevices ? numDevices : 1), 0);
        st = onCpuWorkers(handle, h_inputString, size, numDevices, bound, [&](size_t i, size_t lo, size_t hi, const int *res) {
            int z = 0;                                                           /* the slice's pairs at the slice's own offset, positions counted from the start of the stream */
            for (size_t k = 0; k < hi - lo; k++)
                if (res[k] > 0) { h_matched_result[lo + (size_t)z] = res[k]; h_pos[lo + (size_t)z] = (int)(lo + k); z++; }
            counts[i] = z;
        });
    } else {
        int visible = 0;
        if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) { (void)hipGetLastError(); return PFAC_STATUS_LIB_NOT_EXIST; }
        counts.assign((size_t)(numDevices ? numDevices : visible), 0);
        st = onDevices(handle, size, numDevices, devices, bound, [&](PFAC_context *w, size_t i, size_t lo, size_t hi) {
            return matchHostReduceOnGpu(w, h_inputString + lo, hi - lo, size - lo, lo, h_matched_result + lo, h_pos + lo, &counts[i]);
        });
    }
    if (st != PFAC_STATUS_SUCCESS) return st;
    size_t total = 0;
    for (size_t i = 0; i + 1 < bound.size(); i++) {
        const size_t n = (size_t)counts[i];
        if (n && bound[i] != total) {
            std::memmove(h_matched_result + total, h_matched_result + bound[i], n * sizeof(int));
            std::memmove(h_pos + total, h_pos + bound[i], n * sizeof(int));
        }
        total += n;
    }
    *h_num_matched = (int)total;
    return PFAC_STATUS_SUCCESS;
}

} /* extern "C" */

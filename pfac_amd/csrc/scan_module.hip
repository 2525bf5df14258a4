/*
 * scan_*.hip -- kernel module libpfac_gfx950.so: the PFAC match path for
 * CDNA4 (MI355X).  Hand-written HIP for gfx950 only.
 *
 * Replaces the reference's PFAC_kernel_timeDriven / PFAC_kernel_spaceDriven
 * (PFAC/src/PFAC_kernel.cu:377-458, PFAC/src/PFAC_kernel_spaceDriven.cu:465-558),
 * their host wrappers (:90-244 / :149-348) and the two compaction kernels
 * (PFAC_reduce_kernel.cu, PFAC_reduce_inplace_kernel.cu).  Result contract is identical:
 * d_matched_result[j] = ID of the longest pattern starting at byte j, else 0,
 * every element written.
 *
 * Design (DESIGN.md has the numbers):
 *
 *   The path is HBM-bound integer work: 1 B read + 4 B written per input byte.
 *   The reference walks the automaton from every byte; on MI355X that makes the
 *   CU's address/L1 pipeline (one gathered table line per lane per step), not
 *   HBM, the limit.  pfac_scan_filter is one persistent 1024-thread block per CU
 *   with two kinds of waves:
 *
 *   WRITER waves (3 of 16) only zero-fill: they claim 8 KiB spans of the input in
 *   order -- ONE NARROW MOVING FRONT over the whole grid: granules of 16 spans dealt
 *   round-robin to 2 device counters, so all 256 CUs work inside one 256 KiB window of
 *   the input that sweeps the buffers once --, write the 32 KiB of zeros of the
 *   span with non-temporal 16 B/lane stores, wait until those are in L2 and publish
 *   the span in an LDS ring.  The result stream is 80 % of the traffic and does not
 *   depend on the input; kept out of the scanning waves it neither stalls them nor
 *   is stalled by them.
 *
 *   SCANNING waves take 2 KiB chunks of published spans from an LDS ticket counter:
 *   1. LEVEL 1 (every position, LDS only): lane l tests its 16 positions of each tile
 *      against a blocked two-bit 3-gram Bloom bitmap (two bits of one dword per 3-gram:
 *      one aligned read, two shifts; patterns of 1-2 bytes are folded into it when the
 *      set is compiled).  A miss proves the result is 0.  The chunk is staged in LDS on
 *      the way; the next chunk is prefetched into nine registers the compiler is told
 *      not to use (v119..v127, inline assembly), so that no register copy ever waits
 *      for it.
 *   2. LIST.  The lanes' hits become one list of 16-bit codes (prefix sum of the hit
 *      counts, one divergent loop).  A chunk in which more than half of the positions
 *      hit is not listed: it goes on the launch's dense list, and the tiled kernel
 *      behind this launch walks its positions in place (its dense mode).
 *   3. LEVEL-4 TEST, one hit per lane: the first four bytes against level 4 of the
 *      prefix ladder (+ length-3 bitmap, + exact 2-byte bitmap); survivors stay in the
 *      list, compacted in place.
 *   4. PREFIX LADDER, one candidate per lane: its 20 bytes are cut out of the stage and
 *      the rolling hashes of its prefixes of 6, 8, ..., 20 bytes are tested against ONE
 *      Bloom bitmap in LDS of "stop" and "go on" trie nodes (pfac_context.h: struct
 *      Filter).  A candidate goes to the wave's walk queue {position, 20 input bytes}
 *      only if it follows some pattern until that pattern is alone on its path, and one
 *      level beyond.
 *   5. WALK.  Each lane runs a split-phase walker (two in the compacted-output variant)
 *      over a device-only "chained" table (tables.cpp; used for BOTH perf modes; compact,
 *      breadth first), 16 bytes per slot: one gathered load per edge byte + up to 7 single-
 *      successor bytes; the load of a step is issued in one trip of the loop and
 *      consumed in the next.  A walk starts in a JUMP table keyed by its first four
 *      bytes and restarts in the initial state's bucket if its prefix is not there.  The
 *      input window stays in registers: the entry's bytes (36 in the full-result kernel,
 *      20 in the compacted-output one) end practically every walk without an input load.
 *      "Texture" mode = buffer-resource loads.
 *   6. PATCH.  A non-zero result overwrites its zero, which the writer wave had in
 *      L2 before the chunk was handed out (same CU, same L2: ordered).
 *   The loop has ONE copy of every stage and ONE wait for vector memory: a trip is
 *   wait -> consume the walkers' slots -> refill -> issue the next slots -> (if the
 *   staged chunk is listed and tested) level 1 of the next chunk + prefetch of the
 *   one after it -> list + level-4 test -> ladder batches while the queue has room.
 *
 *   The compacted-output variant (REDUCE) has no zeros to write and no writer waves;
 *   its scanning waves claim chunks from the device counters themselves and append
 *   (id, position) pairs to one list, which four short launches behind the scan put in
 *   position order (bins of positions, rank inside the bin: PairOrder).  So do the
 *   scanning waves of a -DPFAC_WRITERS=0 build, which then issue the zero stores of
 *   their own chunks: there the patch relies on the single in-order vmcnt counter of
 *   gfx9-family hardware (zero store acknowledged before a later load of the same
 *   wave returns).
 *
 *   The scan never checks a bound: the launcher gives it whole chunks that start at
 *   a 16-byte aligned input byte and end at least maxPatternLen + 64 bytes before the
 *   end of the input.  The <= 15 positions in front and the few thousand behind ride
 *   along in the same launch (ScanArgs::endsIn): scanning waves of the first blocks walk
 *   them with bounds (boundedWalk) before they start scanning.  A call is ONE launch of
 *   this kernel -- no memset in front (the last block out leaves the launch counters
 *   zero and publishes the statistics) -- plus a launch of pfac_scan_tiled that looks at
 *   the dense list (empty: it leaves at once).
 *
 *   pfac_scan_tiled (further down) is the kernel of calls below 32 MiB, of PFACX_KERNEL_NAIVE and of
 *   those dense chunks: one position per thread slot, a group of tiles + halo and the hottest
 *   transition rows in LDS, coalesced result lines.  pfac_scan_naive (one thread per byte through the
 *   reference-layout tables, PFACX_KERNEL_REFTABLE) is the reference-shaped baseline and the
 *   independent second implementation the tests cross-check against; it is on no default path.
 *   No MFMA: nothing here is a contraction.
 */
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "scan_*.hip is written for gfx950 (CDNA4): wave64, gfx9 waitcnt semantics, 160 KiB LDS"
#endif
#include <hip/hip_runtime.h>
#include <chrono>

#include <cstdio>
#include <cstring>
#include <mutex>

#include <cstdint>
#include <type_traits>
#include <vector>

#include "pfac_context.h"
#include "scan_common.h"

namespace {

uint32_t clampExtent(size_t bytes) { return bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bytes; }

/* kernel arguments shared by the full-result and the compacted-result paths */
PFAC_status_t fillArgs(const PFAC_context *c, bool hashed, const char *d_input_string, size_t input_size,
                       int *d_matched_result, ScanArgs &a)
{
    if (!c->d_initialRow || !c->d_gram3 || !c->d_ladder || !c->d_final3 || !c->d_shortBits || !c->d_workCounters || !c->d_gram1 || !c->d_prefix4) return PFAC_STATUS_INTERNAL_ERROR;
    if (!c->d_chainSlots || c->chainJumpLog2 <= 0) return PFAC_STATUS_INTERNAL_ERROR;
    /* the reference-layout tables are on the device only while PFACX_KERNEL_REFTABLE is selected (pfac_api.cpp uploads them) */
    if (c->kernelVariant == PFACX_KERNEL_REFTABLE && (hashed ? (!c->d_hashRow || !c->d_hashVal) : !c->d_dense)) return PFAC_STATUS_INTERNAL_ERROR;
    a = ScanArgs{};
    a.in = reinterpret_cast<const unsigned char *>(d_input_string);
    a.out = d_matched_result;
    a.n = a.owned = input_size;
    a.dense = c->d_dense;
    a.hashRow = c->d_hashRow;
    a.hashVal = c->d_hashVal;
    a.denseBytes = clampExtent(c->h_dense.size() * sizeof(int));
    a.hashRowBytes = clampExtent(c->h_hashRow.size() * sizeof(Int2));
    a.hashValBytes = clampExtent(c->h_hashVal.size() * sizeof(Int2));
    a.chainSlots = reinterpret_cast<const u32x4 *>(c->d_chainSlots);
    a.jumpShift = 32u - (uint32_t)c->chainJumpLog2;
    a.extDelta = (uint32_t)(c->numChainSlots / 2);                         /* headers, then as many extension units (tables.cpp) */
    a.jumpBase = (uint32_t)(c->numChainSlots / 2 - (size_t(2) << c->chainJumpLog2));         /* the jump table, then the long jump table */
    a.jumpLongBase = a.jumpBase + (uint32_t)(size_t(1) << c->chainJumpLog2);
    a.rootRow = a.jumpBase - (uint32_t)pfac::kCharSet;
    a.chainBytes = clampExtent(c->numChainSlots * sizeof(pfac::ChainSlot));
    a.initialRow = c->d_initialRow;
    a.gram3 = c->d_gram3;
    a.gram1 = c->d_gram1;
    a.prefix4 = c->d_prefix4;
    a.tail = (c->filter.tail.empty() && c->filter.tailG.empty()) ? nullptr : c->d_tail;       /* a set has the table in one form: LDS slots or device-memory buckets */
    a.log2Tail = c->filter.tail.empty() ? c->filter.log2TailG : c->filter.log2Tail;
    a.shortBits = c->d_shortBits;
    a.ladder = c->d_ladder;
    a.final3 = c->d_final3;
    a.log2Bits = c->filter.log2Bits;
    a.log2BitsLad = c->filter.log2BitsLad;
    a.ladderLast = c->filter.ladderLast;
    a.ladderSalt = c->filter.ladderSalt;
    a.skipCount = c->filter.skipCount;
    for (int k = 0; k < pfac::kSkipTagsMax; k++) a.skipTags[k] = c->filter.skipTags[k];
    a.log2BitsF3 = c->filter.log2BitsF3;
    a.numFinal = c->fa.numPatterns;
    a.maxWalk = (uint32_t)c->fa.maxPatternLen;
    a.work = c->d_workCounters;
    a.hostHint = c->d_modeHint;
    a.denseWord = (uint32_t)pfac::kDenseCountWord;
    a.denseWordOther = (uint32_t)pfac::kDenseCountWordB;
    a.initialState = c->fa.initialState;
    /* the buffer-resource ("texture") path addresses the table with 32-bit byte offsets; the
     * reference fails the texture bind for an oversized table the same way (PFAC_kernel.cu:139-142) */
    if (c->textureMode == PFAC_TEXTURE_ON) {
        const size_t chained = c->numChainSlots * sizeof(pfac::ChainSlot), dense = hashed ? 0 : c->h_dense.size() * sizeof(int);
        if ((chained > dense ? chained : dense) > 0xFFFFFFFFull) return PFAC_STATUS_CUDA_ALLOC_FAILED;
    }
    return PFAC_STATUS_SUCCESS;
}

/* the tiled kernel's table: the narrow chained table (no long slots: tables.cpp) unless the handle's last full-result filter launch found its stream
 * full of near misses (the word the walker choice reads), or PFACX_WALKER_STAGE / _VETO says the caller expects them */
void tiledTable(const PFAC_context *c, ScanArgs &a)
{
    if (c->d_chainNarrow == nullptr || c->numChainNarrow == 0) return;
    const bool nearMisses = (c->h_modeHint != nullptr && *static_cast<volatile const unsigned int *>(c->h_modeHint) != 0) ||
                            c->walker == PFACX_WALKER_STAGE || c->walker == PFACX_WALKER_VETO;
    if (nearMisses) return;
    a.chainSlots = reinterpret_cast<const u32x4 *>(c->d_chainNarrow);
    a.jumpShift = 32u - (uint32_t)c->chainNarrowJumpLog2;
    a.jumpBase = (uint32_t)(c->numChainNarrow - (size_t(1) << c->chainNarrowJumpLog2));
    a.jumpLongBase = a.jumpBase;                       /* (no long jump table: nothing in the narrow table is long) */
    a.rootRow = a.jumpBase - (uint32_t)pfac::kCharSet;
    a.extDelta = 0;
    a.chainBytes = clampExtent(c->numChainNarrow * sizeof(pfac::ChainSlot));
}

/* below this many positions a call takes the tiled kernel alone: ~8 us + what the positions cost instead of the filter
 * kernel's ~19 us floor (filling ~100 KiB of LDS tables per block, the ring of writer and scanning waves).  On the
 * Snort-style stream the two cross between 32 and 64 MiB (tools/small_input_latency.py: 16 MiB 29.6 / 37.7 us, 32 MiB
 * 50.7 / 53.5, 64 MiB 88 / 79; profiles/r04_small_input_latency.txt) */
constexpr size_t kSmallInput = size_t(32) << 20;

/* Launch plan for positions [first, ownEnd) of an input of inputSize readable bytes:
 *   [first, first + mainLen)   filter kernel: whole chunks whose walks stay >= 64 bytes inside the input
 *                              (a walk is at most maxPatternLen deep, a window load reads <= 35 bytes on, the
 *                              prefetch of a chunk the 64 bytes behind it)
 *   [first + mainLen, ownEnd)  bounds-checked walks inside the same launch (ScanArgs::endsIn): the end of the input
 * (`first` is the first 16-byte aligned input byte: scan() and reduceScan() peel the positions in front of it) */
size_t filterLength(const PFAC_context *c, size_t first, size_t ownEnd, size_t inputSize, bool vectorOk, bool reduce = false)
{
    if (!vectorOk || c->kernelVariant == PFACX_KERNEL_NAIVE || c->kernelVariant == PFACX_KERNEL_REFTABLE) return 0;
    if (c->kernelVariant == PFACX_KERNEL_AUTO) {
        if (ownEnd - first < kSmallInput) return 0;        /* filling ~90 KiB of LDS tables per block costs more than scanning this */
        /* the handle's last big launch found most of its stream pattern-dense (short patterns over text, runs of a pattern byte):
         * the filter kernel would list nearly every chunk for the tiled kernel after testing it; the tiled kernel takes the call
         * alone (snort-length set with 1-byte patterns: 133 -> 166 GB/s) and reports when the stream stops being dense */
        /* (full-result calls only: a compacted-output launch neither lists dense chunks nor reports on its stream, so it could never
         * take the hint back -- and its filter kernel has no zeros to lose to the tiled kernel's coalesced lines) */
        if (!reduce && c->h_modeHint != nullptr && static_cast<volatile const unsigned int *>(c->h_modeHint)[1] != 0) return 0;
    }
    const size_t margin = (size_t)c->fa.maxPatternLen + 64 + kWalkHalo;   /* a window load reads up to 35 bytes beyond a walk's deepest byte; the prefetch of a chunk reads the 64 (full-result kernel: kWalkHalo) bytes behind it */
    const size_t safeEnd = inputSize > margin ? inputSize - margin : 0;
    const size_t end = ownEnd < safeEnd ? ownEnd : safeEnd;
    return end > first ? (end - first) / kChunkBytesHost * kChunkBytesHost : 0;
}

/* The vector kernel keeps byte positions in 32 bits: larger inputs are scanned as consecutive windows */
constexpr size_t kMaxLaunchBytes = (size_t(1) << 32) - (size_t(1) << 24);

/* The filter kernel reads the input 16 bytes per lane: it starts at the first 16-byte aligned input byte.  The (at most
 * 15) positions in front of it are walked with bounds, like the end of the input.  The result vector needs no alignment
 * beyond that of an int (its 16-byte stores then straddle lines; a 1 KiB-per-instruction stream does not care). */
size_t headPositions(const unsigned char *in, size_t input_size)
{
    const size_t head = (16u - (reinterpret_cast<uintptr_t>(in) & 15u)) & 15u;
    return head < input_size ? head : input_size;
}

PFAC_status_t scan(PFAC_handle_t handle, char *d_input_string, size_t input_size, int *d_matched_result, bool hashed)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    const PFAC_context *c = handle;
    ScanArgs a;
    const PFAC_status_t st = fillArgs(c, hashed, d_input_string, input_size, d_matched_result, a);
    if (st != PFAC_STATUS_SUCCESS) return st;
    const bool tex = (c->textureMode == PFAC_TEXTURE_ON);
    const bool vectorOk = (reinterpret_cast<uintptr_t>(a.out) & 3u) == 0;       /* an int vector that is not int-aligned: tiled kernel only */
    hipError_t e = hipSuccess;
    const size_t head = vectorOk ? headPositions(a.in, input_size) : 0;
    bool headDone = head == 0;
    for (size_t first = head; first < input_size && e == hipSuccess; first += kMaxLaunchBytes) {
        const size_t ownEnd = input_size - first < kMaxLaunchBytes ? input_size : first + kMaxLaunchBytes;
        const size_t mainLen = filterLength(c, first, ownEnd, input_size, vectorOk);
        ScanArgs part = a;
        part.in = a.in + first;
        part.out = a.out + first;
        if (mainLen) {
            /* room for the list of pattern-dense chunks this launch may leave to the tiled kernel: a grow-only buffer of
             * the handle (the caller holds its lock) */
            const size_t chunks = mainLen / kChunkBytesHost;
            if (handle->denseListEntries < chunks) {
                if (handle->d_denseList) (void)hipFree(handle->d_denseList);
                handle->d_denseList = nullptr;
                handle->denseListEntries = 0;
                if (hipMalloc(reinterpret_cast<void **>(&handle->d_denseList), chunks * sizeof(unsigned int)) != hipSuccess) {
                    (void)hipGetLastError();
                    handle->d_denseList = nullptr;
                    return PFAC_STATUS_CUDA_ALLOC_FAILED;
                }
                handle->denseListEntries = chunks;
            }
            part.denseList = handle->d_denseList;
            part.denseWord = (uint32_t)(handle->denseParity ? pfac::kDenseCountWordB : pfac::kDenseCountWord);
            part.denseWordOther = (uint32_t)(handle->denseParity ? pfac::kDenseCountWord : pfac::kDenseCountWordB);
            handle->denseParity ^= 1u;
            part.n = part.owned = mainLen;
            /* the ends of this window ride along: the positions in front of the first aligned byte (first window only)
             * and what is left behind the last whole chunk */
            const size_t back = headDone ? 0 : head;
            part.endsIn = part.in - back;
            part.endsOut = part.out - back;
            part.endsReadable = input_size - first + back;
            part.endsA0 = 0;
            part.endsA1 = (uint32_t)back;
            part.endsB0 = (uint32_t)(back + mainLen);
            part.endsB1 = (uint32_t)(back + (ownEnd - first));
            headDone = true;
            e = pfacmod::launchFilterKernel(c, part, tex, false);
#if !defined(PFAC_EXP_NO_DENSE_LAUNCH)    /* timing experiment: what the second launch of a call costs (results are wrong if a chunk is dense) */
            if (e == hipSuccess) {
                /* the chunks the filter launch listed as pattern-dense (a launch that finds none leaves at once) */
                ScanArgs rest = part;
                rest.endsIn = nullptr;
                rest.denseIn = part.in;
                rest.denseOut = part.out;
                rest.denseReadable = input_size - first;
                rest.owned = 0;
                rest.n = input_size - first;
                tiledTable(c, rest);
                e = pfacmod::launchTiledKernel(c, rest, tex);
            }
#endif
        } else {
            /* no filter launch (a small call, PFACX_KERNEL_NAIVE / REFTABLE, an odd result pointer): the tiled (or reference-shaped) kernel does it all */
            const size_t back = headDone ? 0 : head;
            ScanArgs rest = part;
            rest.in = part.in - back;
            rest.out = part.out - back;
            rest.owned = ownEnd - first + back;
            rest.n = input_size - first + back;
            headDone = true;
            rest.reportDense = (c->kernelVariant == PFACX_KERNEL_AUTO && vectorOk && ownEnd - first >= kSmallInput) ? 1u : 0u;   /* a big call sent here for its density: say if it still is */
            /* ... and if the set is one that does not fold into chains (a few all-final states: PFAC_context::d_denseFast), through the dense table:
             * a step is one byte either way there, and one gathered dword costs a third of the chained step's instructions (input in which every
             * position matches, 256 MiB: 94 -> 123 GB/s) */
            if (rest.reportDense != 0 && c->d_denseFast != nullptr) {
                rest.dense = c->d_denseFast;
                rest.denseBytes = clampExtent(c->denseFastEntries * sizeof(int));
                e = pfacmod::launchDenseTableKernel(c, tex, rest);
            } else {
                if (c->kernelVariant != PFACX_KERNEL_REFTABLE) tiledTable(c, rest);
                e = pfacmod::launchSimpleKernel(c, hashed, tex, rest);
            }
        }
    }
    if (e == hipSuccess && !headDone) {                 /* the whole input is in front of the first aligned byte */
        ScanArgs part = a;
        part.owned = head;
        part.n = input_size;
        if (c->kernelVariant != PFACX_KERNEL_REFTABLE) tiledTable(c, part);
        e = pfacmod::launchSimpleKernel(c, hashed, tex, part);
    }
    return e == hipSuccess ? PFAC_STATUS_SUCCESS : PFAC_STATUS_INTERNAL_ERROR;
}

#include "scan_order.inc"
/*
 * Compacted output (ref PFAC_reduce_kernel / PFAC_reduce_inplace_kernel, PFAC_reduce_kernel.cu:172-295,
 * PFAC_reduce_inplace_kernel.cu:155-323): the first *h_num_matched entries of d_match_result / d_pos
 * receive the non-zero results and their positions in ascending position order.
 *
 * Same kernel as the full-result path with REDUCE = true: no zero stores (the 4 B/byte output wall
 * is gone, traffic is ~1 B per input byte), finished walkers append (id, position) through one
 * device counter.  The ends of the input (ScanArgs::endsIn) are walked with bounds by the first blocks of the same
 * launch and join the list through the same counter (a small input: the tiled kernel appends).  The list is then put in position
 * order (PairOrder) by launches queued behind the scan; the host reads the count once, at the end (synchronous, like
 * the reference's call).
 * The reference needs a block-local compaction, a Thrust scan and a second gather kernel
 * (PFAC_reduce_kernel.cu:417-457) because it has no prefilter: every thread owns a result.
 */
PFAC_status_t reduceScan(PFAC_handle_t handle, int *d_input_string, int input_size, int *d_match_result, int *d_pos,
                         int *h_num_matched, int *h_match_result, int *h_pos, bool hashed)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!d_input_string || !d_match_result || !d_pos || !h_num_matched || input_size <= 0) return PFAC_STATUS_INVALID_PARAMETER;
    const PFAC_context *c = handle;
    const size_t n = (size_t)input_size;
    ScanArgs a;
    PFAC_status_t st = fillArgs(c, hashed, reinterpret_cast<const char *>(d_input_string), n, d_match_result, a);
    if (st != PFAC_STATUS_SUCCESS) return st;
    const bool tex = (c->textureMode == PFAC_TEXTURE_ON);
    const bool ordered = !c->reduceUnordered;              /* PFAC_matchFromHost scatters the pairs: any order */

    /* the handle's scratch: the counters of this call (pairs, pairs per position bin), room to order the pairs through */
#ifndef PFAC_REDUCE_TRACE
#define PFAC_REDUCE_TRACE 0                    /* measurement build: host-side timeline of a compacted-output call on stderr */
#endif
#if PFAC_REDUCE_TRACE
    const auto tr0 = std::chrono::steady_clock::now();
    auto trUs = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count(); };
    double trPlan = 0, trScan = 0, trOrder = 0, trDone = 0;
#endif
    PairOrder order;
    const size_t expected = n / 128 > 65536 ? n / 128 : 65536;       /* room for one match per 128 bytes before the first call has been seen */
    st = order.plan(handle, n, expected, d_match_result, d_pos, handle->orderParity);
    if (st != PFAC_STATUS_SUCCESS) return st;
    /* an ordered call cleans up behind itself (scan_order.inc) and hands the number of pairs over in mapped host memory: no memset in
     * front of the scan when the previous call has left this very layout clean, no device-to-host copy behind the last launch */
    const bool tidy = ordered && order.o.hostCount != nullptr && c->h_modeHint != nullptr;
    const bool clean = tidy && handle->orderCleanBase == order.o.counts && handle->orderCleanBytes == order.counterBytes;
    handle->orderCleanBase = nullptr;                      /* until this call has ended well */
    /* (PFACX_setKernelTiming: the memset stays -- the event in front of the scan kernel would otherwise be recorded by an idle queue, and the
     * kernel's time would include the queue's wake-up) */
    if ((!clean || c->kernelTiming) && order.clearCounters() != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
#if PFAC_REDUCE_TRACE
    trPlan = trUs();
#endif
    const size_t head = headPositions(a.in, n);
    const size_t mainLen = filterLength(c, head, n, n, true, /*reduce=*/true);
    a.reducePos = d_pos;
    a.reduceCount = order.o.count;
    if (mainLen) {
        ScanArgs part = a;
        part.in = a.in + head;
        part.n = part.owned = mainLen;
        part.reduceBase = (unsigned int)head;
        /* the ends of the input ride along (ScanArgs::endsIn): their matches join the list through the same counter */
        part.endsIn = a.in;
        part.endsReadable = n;
        part.endsA0 = 0;
        part.endsA1 = (uint32_t)head;
        part.endsB0 = (uint32_t)(head + mainLen);
        part.endsB1 = (uint32_t)n;
        if (pfacmod::launchFilterKernel(c, part, tex, true) != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
    } else {
        /* a small input (or PFACX_KERNEL_NAIVE / REFTABLE): positions [0, n) through the tiled (reference-shaped) kernel, which appends its matches to the list */
        ScanArgs part = a;
        part.owned = n;
        part.reduceBase = 0;
        if (c->kernelVariant != PFACX_KERNEL_REFTABLE) tiledTable(c, part);
        if (pfacmod::launchSimpleKernel(c, hashed, tex, part) != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
    }
#if PFAC_REDUCE_TRACE
    trScan = trUs();
#endif
    unsigned int count = 0;
    if (tidy) {
        volatile unsigned int *hostCount = c->h_modeHint + pfac::kHostPairCountWord, *hostDone = hostCount + 1;
        handle->orderSeq = handle->orderSeq + 1u ? handle->orderSeq + 1u : 1u;
        order.o.seq = handle->orderSeq;
        *hostCount = 0xFFFFFFFFu;
        if (order.order(c) != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
#if PFAC_REDUCE_TRACE
        trOrder = trUs();
#endif
        /* the last launch writes the call's number into host memory: polled for a while (the call is a millisecond of GPU work per GiB) */
        const auto t0 = std::chrono::steady_clock::now();
        bool through = false;
        for (unsigned int spins = 0; !(through = __atomic_load_n(const_cast<unsigned int *>(hostDone), __ATOMIC_ACQUIRE) == order.o.seq); spins++) {
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
        if (!through && hipStreamSynchronize(0) != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
        count = *hostCount;
#if PFAC_REDUCE_TRACE
        trDone = trUs();
        fprintf(stderr, "PFAC_REDUCE_TRACE us: planned %.1f, scan queued %.1f, ordering queued %.1f, done %.1f (clean %d, polled %d)\n", trPlan, trScan, trOrder, trDone, (int)clean, (int)through);
#endif
    } else {
        if (ordered && order.order(c) != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
        if (hipMemcpy(&count, order.o.count, sizeof(count), hipMemcpyDeviceToHost) != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
    }
    if (count > (unsigned int)input_size) return PFAC_STATUS_INTERNAL_ERROR;
    if (ordered && count > order.o.capacity) {             /* more pairs than the scratch held: the launches left at once */
        st = order.plan(handle, n, count, d_match_result, d_pos, handle->orderParity);
        if (st != PFAC_STATUS_SUCCESS) return st;
        if (order.clearCounters() != hipSuccess || hipMemcpyAsync(order.o.count, &count, sizeof(count), hipMemcpyHostToDevice, 0) != hipSuccess ||
            order.order(c) != hipSuccess || hipStreamSynchronize(0) != hipSuccess)      /* `count` is read by that copy */
            return PFAC_STATUS_INTERNAL_ERROR;
    } else if (tidy) {
        handle->orderParity ^= 1u;                         /* the counters this call has just left zero */
        handle->orderCleanBase = order.o.counts;
        handle->orderCleanBytes = order.counterBytes;
    }
    *h_num_matched = (int)count;
    if (count && h_match_result && hipMemcpy(h_match_result, d_match_result, count * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
        return PFAC_STATUS_INTERNAL_ERROR;
    if (count && h_pos && hipMemcpy(h_pos, d_pos, count * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
        return PFAC_STATUS_INTERNAL_ERROR;
    return PFAC_STATUS_SUCCESS;
}

/* ---------------------------------------------------------- stream probe (measurement only) */

/* The traffic shape of the match path with nothing else in it: every wave reads 1 KiB of the input and writes 4 KiB
 * of zeros (non-temporal), small blocks in dispatch order.  bench.py runs it on the very buffers it has just timed
 * the scan on and reports it next to the scan ("what this part sustains for 1 B read : 4 B written"). */
__global__ __launch_bounds__(256) void pfac_stream_1r4w(const u32x4 *in, i32x4 *out, unsigned int *sink)
{
    const size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const u32x4 v = in[tile * 64 + lane];
    const i32x4 z = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) __builtin_nontemporal_store(z, &out[tile * 256 + k * 64 + lane]);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = v.x;      /* keeps the load */
}

} // namespace

extern "C" {

/* compile-time shape of this module, for the bench record */
#define PFAC_STR2(x) #x
#define PFAC_STR(x) PFAC_STR2(x)
const char *PFACX_buildInfo(void)
{
    return "gfx950 block=" PFAC_STR(PFAC_BLOCK_THREADS) " writers=" PFAC_STR(PFAC_WRITERS) " walk_sets=" PFAC_STR(PFAC_WALK_SETS_FULL) "/" PFAC_STR(PFAC_WALK_SETS) " queue=" PFAC_STR(PFAC_QUEUE_CAP)
           " list=" PFAC_STR(PFAC_LIST_CAP) " span_log2=" PFAC_STR(PFAC_SPAN_LOG2) " front_log2=" PFAC_STR(PFAC_FRONT_LOG2) " parts=" PFAC_STR(PFAC_WORK_PARTS)
           " refill_min=" PFAC_STR(PFAC_REFILL_MIN) " ablate=" PFAC_STR(PFAC_ABLATE) " timing=" PFAC_STR(PFAC_TIMING);
}

/* average milliseconds of `launches` back-to-back launches of pfac_stream_1r4w over the first n (a multiple of 4096)
 * bytes of d_in, 4 n bytes of d_out are overwritten with zeros; < 0: a HIP error */
double PFACX_streamProbe(const void *d_in, void *d_out, size_t n, int launches)
{
    if (!d_in || !d_out || n < 4096 || launches < 1) return -1.0;
    unsigned int *sink = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    double ms = -1.0;
    if (hipMalloc(reinterpret_cast<void **>(&sink), sizeof(unsigned int)) == hipSuccess && hipEventCreate(&a) == hipSuccess &&
        hipEventCreate(&b) == hipSuccess) {
        const unsigned blocks = (unsigned)(n / 4096);
        for (int r = 0; r < 3; r++)
            hipLaunchKernelGGL(pfac_stream_1r4w, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const u32x4 *>(d_in), reinterpret_cast<i32x4 *>(d_out), sink);
        (void)hipEventRecord(a, 0);
        for (int r = 0; r < launches; r++)
            hipLaunchKernelGGL(pfac_stream_1r4w, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const u32x4 *>(d_in), reinterpret_cast<i32x4 *>(d_out), sink);
        (void)hipEventRecord(b, 0);
        float t = 0;
        if (hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&t, a, b) == hipSuccess && hipGetLastError() == hipSuccess) ms = (double)t / launches;
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (sink) (void)hipFree(sink);
    return ms;
}

PFAC_status_t PFAC_kernel_timeDriven_warpper(PFAC_handle_t handle, char *d_input_string, size_t input_size,
                                             int *d_matched_result)
{
    return scan(handle, d_input_string, input_size, d_matched_result, false);
}

PFAC_status_t PFAC_kernel_spaceDriven_warpper(PFAC_handle_t handle, char *d_input_string, size_t input_size,
                                              int *d_matched_result)
{
    return scan(handle, d_input_string, input_size, d_matched_result, true);
}

PFAC_status_t PFAC_reduce_kernel(PFAC_handle_t handle, int *d_input_string, int input_size, int *d_match_result,
                                 int *d_pos, int *h_num_matched, int *h_match_result, int *h_pos)
{
    return reduceScan(handle, d_input_string, input_size, d_match_result, d_pos, h_num_matched, h_match_result, h_pos, false);
}

PFAC_status_t PFAC_reduce_inplace_kernel(PFAC_handle_t handle, int *d_input_string, int input_size, int *d_match_result,
                                         int *d_pos, int *h_num_matched, int *h_match_result, int *h_pos)
{
    return reduceScan(handle, d_input_string, input_size, d_match_result, d_pos, h_num_matched, h_match_result, h_pos, true);
}

} /* extern "C" */

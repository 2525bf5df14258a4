/*
 * scan_filter.hip -- pfac_scan_filter: the product kernel of the PFAC match path for CDNA4 (MI355X), with both walkers and the
 * compacted-output variant, and its launcher.
 * (part of the kernel module libpfac_gfx950.so: see scan_common.h and scan_module.hip)
 */
#include "scan_common.h"

namespace {

/* --------------------------------------------------------- filter kernel */

#ifndef PFAC_ABLATE
#define PFAC_ABLATE 0                         /* timing experiments only (tools/ab.sh): 1 = stream + level 1, 2 = no walks */
#endif
#ifndef PFAC_STATS
#define PFAC_STATS 0                          /* -DPFAC_STATS=1: per-block counters printed at kernel end (PFAC_STATS build) */
#endif
/* A scanning wave works on one CHUNK of two 1 KiB tiles at a time: the chunk is staged in LDS (+ the 32 bytes
 * behind it), every lane's level-1 hits go to a per-wave list of 16-bit codes, and 64 list entries at a time
 * go through the level-4 test and the prefix ladder; what is left is cut out of the stage and appended to the walk
 * queue -- one entry per lane. */
constexpr int kGroupTiles = pfac::kChunkTiles;
constexpr int kGroupBytes = kGroupTiles * kTileBytes;
constexpr int kStageWords = (kGroupBytes + 48) / 4;      /* the chunk + the 48 bytes behind it: an entry is cut up to 36 bytes deep */
#ifndef PFAC_REFILL_MIN
#define PFAC_REFILL_MIN 16                     /* queue entries are handed out only when at least this many lanes of a walk set are idle:
                                                * a refill costs the whole wave ~40 instructions however few lanes it fills
                                                * (C5 1.94 -> 1.82 ms, C3 -1 %; profiles/r02_ab_refill.txt) */
#endif
#ifndef PFAC_REFILL_BATCH
#define PFAC_REFILL_BATCH 1                    /* ... and only when at least this many entries are queued (or the wave has no more chunks to filter) */
#endif
#ifndef PFAC_TIMING
#define PFAC_TIMING 0
#endif
#ifndef PFAC_LIST_CAP
#define PFAC_LIST_CAP 128
#endif
constexpr uint32_t kListCap = PFAC_LIST_CAP;  /* 16-bit hit codes per wave; more level-1 hits in one chunk take another round */
#ifndef PFAC_REDUCE_PARTS
#define PFAC_REDUCE_PARTS 32                    /* claim counters of the compacted-output kernel (its waves claim granules of chunks themselves) */
#endif
/* The compacted-output kernel tests no ladder level behind depth 4: it is bound by instruction issue, not by the memory system,
 * and there a walk is cheaper than the levels that would spare it (round 3: C3 0.97 / 0.96 / 0.93 / 0.90 ms per call with 8 / 4 /
 * 2 / 0 levels, C5 1.37 / 1.28 / 1.22 / 1.16).  Its LDS holds gram1 and prefix4 (pfac_context.h) instead of gram3 and the ladder. */
#ifndef PFAC_APPEND_MIN
#define PFAC_APPEND_MIN 48                     /* a ladder / append batch that the walk queue's room cuts short takes at least this many candidates (or waits
                                                * for room): on walk-bound input the queue is always nearly full and batches of 16 cost as many instructions as
                                                * full ones (C5 1.559 / 1.545 / 1.532 ms with 16 / 32 / 48; 64 = 48; C3 unchanged) */
#endif
constexpr uint32_t kAppendMin = PFAC_APPEND_MIN;
#ifndef PFAC_MERGE_MIN
#define PFAC_MERGE_MIN 48                      /* tested candidates left in the list while the chunk has more hits: fewer than this wait for the next list round */
#endif
constexpr uint32_t kMergeMin = PFAC_MERGE_MIN;
static_assert(kMergeMin <= 64 && kMergeMin < kListCap, "left-over candidates are moved to the list's head one per lane");
#ifndef PFAC_PATCH_STAGED
#define PFAC_PATCH_STAGED 0                    /* full-result kernel: 1 = finished matches are staged per wave in LDS and stored kReduceCap at a time */
#endif
constexpr bool kStagedPatch = PFAC_PATCH_STAGED != 0;
constexpr uint32_t kReduceCap = 16;           /* (position, id) pairs staged per wave in the REDUCE variant (a ballot with more goes out directly) */
constexpr int kReduceScanners = kWavesPerBlock;       /* ... and no writer waves: every wave scans, with half the walk queue each (LDS) */
constexpr uint32_t kReduceQueueCap = kQueueCap;
/* In-order hand-out of the input: -1 = the input is cut into kWorkParts contiguous parts, one
 * counter each; G >= 0 = one moving front: granules of 2^G pieces are dealt round-robin to the parts, so
 * all parts work inside one window of parts << G pieces that sweeps the input once. */
#ifndef PFAC_FRONT_LOG2
#define PFAC_FRONT_LOG2 4
#endif

/* Zero-fill by dedicated WRITER waves (full-result kernel only).  The API writes 4 bytes per input byte, almost
 * all zero, and that stream does not depend on the input.  Issued by the scanning waves themselves it ties
 * their progress to the store path: a wave that waits for room in the store queue is not filtering, and with 4
 * waves per SIMD there is little else to run.  So the last kWriters waves of a block do nothing but zero-fill:
 * a writer claims the next SPAN (kSpanChunks chunks) of the block's part, fills its results with zeros, waits
 * until the stores have reached L2 and publishes the span in an LDS ring; the other waves take chunks of
 * published spans from an LDS ticket counter and only ever store matches, on top of zeros that are already in
 * L2 (same CU, same L2: ordered).  Writers run at most kRunAhead spans ahead of the scanners. */
#ifndef PFAC_WRITERS
#define PFAC_WRITERS 3
#endif
#ifndef PFAC_SPAN_LOG2
#define PFAC_SPAN_LOG2 2
#endif
constexpr int kSpanLog2 = PFAC_SPAN_LOG2;
constexpr uint32_t kSpanChunks = 1u << kSpanLog2;      /* chunks per span (4 chunks = 8 KiB of input, 32 KiB of results) */
/* writers run up to kRunAhead spans (>= 128 KiB of input) ahead of the tickets handed out; a scanner holds at
 * most 2 tickets it has not resolved yet, so a ring slot is reused only kRing - kRunAhead >= 32 tickets later */
#ifndef PFAC_RUN_AHEAD
#define PFAC_RUN_AHEAD ((64u >> kSpanLog2) > 4u ? (64u >> kSpanLog2) : 4u)
#endif
constexpr uint32_t kRunAhead = PFAC_RUN_AHEAD;
constexpr uint32_t kRing = 2 * kRunAhead;
static_assert((kRing - kRunAhead) * kSpanChunks >= 2 * 16 + kSpanChunks, "ring slack covers the unresolved tickets of 16 waves");
constexpr uint32_t kEnd = 0xFFFFFFFFu;
struct Control {                                         /* LDS, one per block */
    uint32_t popCount;                                   /* tickets handed to scanners (chunk number in ring order) */
    uint32_t pubCount;                                   /* spans published, in order                               */
    uint32_t claimTurn;                                  /* next block-local span number allowed to claim           */
    uint32_t endSpan;                                    /* first block-local span number past the part's end       */
    uint32_t ring[kRing];                                /* span ids of the published spans                         */
};
constexpr int kControlWords = (sizeof(Control) / 4 + 3) / 4 * 4;

__device__ __forceinline__ uint32_t ldsLoad(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ldsStore(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }


/* minimum over the 64 lanes (same DPP ladder; lanes without a source keep their own value) */
__device__ __forceinline__ uint32_t waveMin(uint32_t v)
{
    auto step = [](uint32_t x, uint32_t y) { return y < x ? y : x; };
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

/* vector registers the compiler may use in the filter kernel, HALVED (on gfx90a and later the attribute counts a
 * unified VGPR + AGPR budget of twice its value; the kernel uses no AGPRs): v0..v117.  v119..v127 hold the chunk
 * in flight (prefetchChunk). */
#ifndef PFAC_COMPILER_VGPRS
#define PFAC_COMPILER_VGPRS 59                  /* (-DPFAC_COMPILER_VGPRS=48 is what tests/test_kernel_isa.py breaks the contract with: `make` then fails) */
#endif
constexpr int kCompilerVgprs = PFAC_COMPILER_VGPRS;

/* a.n is a whole number of chunks (>= 1) and at least maxPatternLen + 64 readable input bytes follow it */
template <bool TEX, bool HAS_SHORT, bool REDUCE, int kWalkSets, bool STAGE, int VETO = 0>
__global__ __launch_bounds__(kBlockThreads, PFAC_MIN_WAVES_PER_SIMD) __attribute__((amdgpu_num_vgpr(kCompilerVgprs)))
void pfac_scan_filter(ScanArgs a)
{
    constexpr int kTilesPerIter = kGroupTiles;
    constexpr int kChunkBytes = kTilesPerIter * kTileBytes;    /* input bytes a wave stages at a time */
    using WCtx = ChainCtx<TEX>;
    constexpr uint32_t kEntry = (REDUCE || (!STAGE && VETO == 2)) ? kEntryBytes : kEntryBytesFull;      /* (VETO = 2: see kVetoG below) */
    constexpr bool kWideEntry = kEntry > kEntryBytes;                                                    /* entries carry bytes 20..35 too (queueC) */
    /* full-result kernel: walks read their input from the wave's two staged chunks (StageLane), a queue entry is {buffer, offset};
     * compacted-output kernel (16 scanning waves, no LDS to spare): the input travels with the entry and lives in registers */
    constexpr bool kStageWalk = !REDUCE && STAGE;
    /* VETO: the register-window walker behind a deeper prefilter -- ladder levels behind the 20th byte and the tail table (pfac::Filter): what a pattern
     * set of a few thousand patterns gets (its tables leave the LDS for it); a stop of the ladder is put to the table before it becomes a walk.  Its
     * walker fetches the extension unit of a wide bucket's slot with the header once its wave has met long slots (what is left to walk are patterns
     * that end within a byte of where the candidate left them: long slots all the way) */
    constexpr bool kVeto = !REDUCE && !STAGE && VETO != 0;
    /* VETO = 2 (round 6): the same kernel for a set whose tail table does not fit the LDS (Snort-scale: the bitmaps take it, and its thin stops outnumber the
     * LDS table's slots): the table lies in device memory (pfac::Filter::tailG), a batch of the ladder with kTailAskMin or more stopped candidates asks it
     * with ONE gathered 16-byte load per candidate, issued behind the batches of a trip and looked at at the top of the next one (tailResolve).  The lines a
     * near-miss stream asks for stay in L2: it meets the same few hundred stop nodes over and over.  Text does not repay the round trip: launchChained gives
     * such a set this kernel only while the handle's launches report near misses.
     * What is left to walk behind this veto is little (4 M walks per GiB of the near-miss stream over the 31 000-pattern set), so the kernel's walker is the
     * compacted-output kernel's: 20-byte queue entries, a five-dword window (a walk deeper than that re-fetches its window: rare enough now) -- four registers
     * less, which the answers on their way need, and 16 bytes per queue entry less, in whose place the wave's LIST lies with 256 codes instead of 128: the
     * alphanumeric near-miss stream has ~127 level-1 hits per chunk under that set, and every second chunk needed a second list round -- a trip of its own. */
    constexpr bool kVetoG = kVeto && VETO == 2;
    constexpr uint32_t kListCapK = kVetoG ? 2u * kListCap : kListCap;
    static_assert(kListCapK * 2 <= kQueueCap * 16 || !kVetoG, "VETO = 2: the list lies in the place of the queue's bytes 20..35");
#ifndef PFAC_TAIL_ASK_MIN
#define PFAC_TAIL_ASK_MIN 8
#endif
    constexpr uint32_t kTailAskMin = PFAC_TAIL_ASK_MIN;
    /* (VETO = 2 keeps six more registers alive across the loop's back edge -- a batch's answers from the tail table -- and leaves the unit speculation,
     * four registers, to VETO = 1: its walker fetches a long slot's unit on the spot, like the plain window walker) */
    using WLane = std::conditional_t<kStageWalk, StageLane<TEX>, ChainLane<TEX, kEntry, kVeto && VETO == 1>>;
    constexpr int kStageWordsK = kStageWalk ? (int)(kWalkStageBytes / 4) : kStageWords;     /* words of one stage buffer */
    constexpr int kStageBufs = kStageWalk ? 2 : 1;
    constexpr int kHaloDwords = kStageWalk ? (int)(kWalkHalo / 4) : 12;                      /* dwords behind the chunk that are staged with it */
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int words3 = 1 << (a.log2Bits - 5), wordsLad = 1 << (a.log2BitsLad - 5), wordsF3 = 1 << (a.log2BitsF3 - 5);
    uint32_t *sGram3 = reinterpret_cast<uint32_t *>(smem);
    uint32_t *sLadder = sGram3 + kLadderLdsOffset / 4;          /* at a compile-time address whatever the size of the level-1 bitmap: a ladder probe's
                                                                  ds_read takes it as its immediate offset */
    uint32_t *sFinal3 = REDUCE ? sGram3 + (kGram1LdsOffset + kGram1LdsBytes) / 4 : sLadder + wordsLad;
    uint32_t *sShort = sFinal3 + wordsF3;
    constexpr int kWriters = REDUCE ? 0 : PFAC_WRITERS;             /* the compacted-output variant has no zeros to write */
    constexpr int kScanners = REDUCE ? kReduceScanners : kWavesPerBlock - kWriters;
    Control *ctl = reinterpret_cast<Control *>(sShort + (HAS_SHORT ? 2048 : 0));
    uint32_t *sQueueAll = reinterpret_cast<uint32_t *>(ctl) + kControlWords;           /* 16-byte aligned */
    constexpr uint32_t kQCap = REDUCE ? kReduceQueueCap : kQueueCap;
    uint32_t *sQueueBAll = sQueueAll + kScanners * kQCap * (kStageWalk ? 1 : 4);   /* ... second part of the entries: input bytes 12..19 (kStageWalk: an entry is one word) */
    uint32_t *sQueueCAll = sQueueBAll + (kStageWalk ? 0 : kScanners * kQCap * 2);  /* ... register-window walkers of a full-result build: input bytes 20..35 */
    uint32_t *sStageAll = sQueueCAll + ((REDUCE || kStageWalk) ? 0 : kScanners * kQCap * 4);      /* (VETO = 2 keeps the room: its list lies there) */   /* per scanning wave: the chunk being filtered + the bytes behind it (kStageWalk: and the chunk before it) */
    uint32_t *sListAll = sStageAll + kScanners * kStageWordsK * kStageBufs;   /* per scanning wave: 16-bit codes of the chunk's level-1 hits */
    uint32_t *sReduceAll = sListAll + kScanners * (kListCap / 2);        /* REDUCE only: per-wave staging of (position, id) */
    uint32_t *sDenseAll = sReduceAll + ((REDUCE || kStagedPatch) ? kScanners * 2 * kReduceCap : 0);   /* full-result kernel: per-wave staging of dense chunk numbers */
    uint32_t *sHotAll = sDenseAll + (REDUCE ? 0 : kScanners * (int)kDenseStage);                         /* kStageWalk: the first a.hotSlots slot headers of the chained table */

    const int tid = threadIdx.x;
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();   /* the level-1 bitmap is addressed by number: sGram3 must sit at LDS address 0 */
    {   /* fill the LDS tables once per (persistent) block, 16 B per lane */
        auto copy16 = [&](uint32_t *dst, const void *src, int words) {
            const u32x4 *g = reinterpret_cast<const u32x4 *>(src);
            u32x4 *s = reinterpret_cast<u32x4 *>(dst);
            for (int i = tid; i < words / 4; i += kBlockThreads) s[i] = g[i];
        };
        if (REDUCE) {
            copy16(sGram3, a.prefix4, (int)(kPrefix4LdsBytes / 4));
            copy16(sGram3 + kGram1LdsOffset / 4, a.gram1, (int)(kGram1LdsBytes / 4));
        } else {
            copy16(sGram3, a.gram3, words3);
            copy16(sLadder, a.ladder, wordsLad);
        }
        copy16(sFinal3, a.final3, wordsF3);
        if (HAS_SHORT) copy16(sShort, a.shortBits, 2048);
        if constexpr (kStageWalk) copy16(sHotAll, a.chainSlots, (int)a.hotSlots * 4);
        if constexpr (kVeto && !kVetoG) { if (a.tail != nullptr) copy16(sHotAll, a.tail, 3 << a.log2Tail); }      /* the tail table: behind everything else */
        if (tid < kControlWords) reinterpret_cast<uint32_t *>(ctl)[tid] = (tid == (int)(offsetof(Control, endSpan) / 4)) ? kEnd : 0u;      /* endSpan = none yet */
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   /* wave-uniform by construction: keep it (and what derives from it) scalar */
    /* ring of {byte position (32-bit), input bytes pos..pos+19} that passed level 1 and the prefix ladder, kept as a 16-byte
     * and an 8-byte array: the twenty bytes carry practically every walk to its end without a single input load
     * (gathered loads are the scarce resource) */
    u32x4 *queue = reinterpret_cast<u32x4 *>(sQueueAll) + wave * kQCap;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 *queueB = reinterpret_cast<u32x2 *>(sQueueBAll) + wave * kQCap;
    u32x4 *queueC = reinterpret_cast<u32x4 *>(sQueueCAll) + wave * kQCap;
    uint32_t *const stage0 = sStageAll + wave * (kStageWordsK * kStageBufs);
    uint32_t *stage = stage0;                                  /* the buffer of the chunk being filtered */
    uint32_t *queue32 = sQueueAll + wave * kQCap;              /* kStageWalk: entries {buffer << 31 | offset in the chunk} */
    /* kStageWalk: the wave's two stage buffers.  `cur` holds the chunk being filtered, the other one the chunk before it, whose
     * walks may still be queued or under way: it is overwritten only when they are through (qEnd: the queue counter behind
     * the last entry of the chunk staged in each buffer). */
    StageView view;
    view.addr[0] = (uint32_t)(reinterpret_cast<unsigned char *>(stage0) - smem);
    view.addr[1] = view.addr[0] + (uint32_t)kStageWordsK * 4u;
    view.base[0] = view.base[1] = 0;
    uint32_t cur = 0, qEnd[2] = {0, 0};
    /* kStageWalk: the wave's stream mode (StageLane): text = entries carry their bytes (in the second buffer's place), stage =
     * two staged chunks.  deepRecent: walks that ran off their LDS bytes lately */
    /* a launch starts in the mode most waves of the handle's previous launch ended in (a stream rarely changes its nature
     * between two calls; a wave that guesses wrong switches after a few chunks) */
    bool modeStage = kStageWalk && __builtin_amdgcn_readfirstlane((int)a.work[pfac::kModeHintWord]) != 0;
    uint32_t deepRecent = 0, stageHold = modeStage ? 8u : 0u;
    uint16_t *list = kVetoG ? reinterpret_cast<uint16_t *>(sQueueCAll + wave * kQCap * 4) : reinterpret_cast<uint16_t *>(sListAll + wave * (kListCap / 2));
    const uint32_t n = (uint32_t)a.n;               /* < 2^32: the launcher splits larger inputs */
    const Lds lds{sGram3, sLadder, sFinal3, sShort,
                  ((1u << ((uint32_t)a.log2Bits - 5u)) - 1u) << 2 /* product's high half -> byte address of the level-1 dword */, 32u - (uint32_t)a.log2BitsLad, 32u - (uint32_t)a.log2BitsF3};
    /* The prefix ladder's bitmap is blocked (pfac_context.h: ladderWord): ALL bits of a node hash h lie in one dword -- bits 18.. of h pick it, its byte
     * address is one SDWA AND of h's high half with (dwords - 1) << 2 --, and 5-bit fields of h number the bits: S = [3..7] and [8..12], G = [13..17], the
     * second G bit of depth 4 = [0..4].  A level is ONE LDS read and no multiplication (rounds 3 - 5: three reads at three hashed places and two 32-bit
     * multiplications -- a third of a level's instructions).  The results carry the bit in bit 0; the bits above it are garbage. */
#if PFAC_LADDER_BLOCKED
    uint32_t vLadMask;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vLadMask) : "s"(((1u << ((uint32_t)a.log2BitsLad - 5u)) - 1u) << 2));
    auto ladWord = [&](uint32_t h) -> uint32_t {
        uint32_t addr;
        asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(addr) : "v"(h), "v"(vLadMask));
        return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(addr + kLadderLdsOffset);
    };
    auto ladStop = [](uint32_t w, uint32_t h) -> uint32_t { return (w >> ((h >> 3) & 31u)) & (w >> ((h >> 8) & 31u)); };
    auto ladGoOn = [](uint32_t w, uint32_t h) -> uint32_t { return w >> ((h >> 13) & 31u); };
    auto ladGoOn2 = [](uint32_t w, uint32_t h) -> uint32_t { return w >> (h & 31u); };
#else       /* measurement builds (-DPFAC_LADDER_BLOCKED=0, host library too): the bitmap of rounds 3 - 5, a probe per bit at a place of its own */
    auto ladProbe = [&](uint32_t v) -> uint32_t {
        const uint32_t idx = v >> lds.shiftLad;
        return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(((idx >> 3) & ~3u) + kLadderLdsOffset) >> (idx & 31u);
    };
    auto ladWord = [](uint32_t h) -> uint32_t { return h; };
    auto ladStop = [&](uint32_t, uint32_t h) -> uint32_t { return ladProbe(h) & ladProbe(h * pfac::kLadMulS); };
    auto ladGoOn = [&](uint32_t, uint32_t h) -> uint32_t { return ladProbe(h * pfac::kLadMulG); };
    auto ladGoOn2 = [&](uint32_t, uint32_t h) -> uint32_t { return ladProbe(h * pfac::kLadMulG2); };
#endif
    WCtx wctx(a);
    if constexpr (kStageWalk) {
        wctx.hotAddr = (uint32_t)(reinterpret_cast<unsigned char *>(sHotAll) - smem);
        wctx.hotSlots = a.hotSlots;
    }
    WLane walk[kWalkSets];
    bool alive[kWalkSets];
#pragma unroll
    for (int s = 0; s < kWalkSets; s++) alive[s] = false;
    /* Ring-queue counters (wave-uniform, monotonically increasing; index = counter & (cap-1)):
     *   [qh, qv)  passed level 1 and the ladder, waiting for a walker lane */
    uint32_t qh = 0, qv = 0;
    constexpr uint32_t kMask = kQCap - 1;
    /* the counters are wave-uniform; saying so keeps them (and every branch on them) on the scalar unit */
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    /* always-on counters of this wave (scalar adds), summed into a.work[kStatsWord..] at kernel end */
    uint32_t stRounds = 0, stLaneSteps = 0, stStarts = 0, stHits = 0, stCand = 0;
#if PFAC_STATS
    uint32_t stFullRounds = 0, stSlotGathers = 0, stWinLoads = 0, stStartDead = 0;
#endif

    /* Reporting a finished walk.  With writer waves the zeros of a chunk are in L2 before the chunk is handed out.
     * Without them (-DPFAC_WRITERS=0) zero stores and walker loads of one wave complete in issue order (a single
     * in-order vmcnt counter on gfx9-family hardware), and a walk ends in consume(), where it has just consumed
     * loads issued behind its chunk's zero stores, so its patch lands on top of the zero.
     * REDUCE: results are staged per wave in LDS and flushed with one atomic per kReduceCap pairs
     * (a single device counter saturates at ~90 increments/us; pattern-dense input has 10^5..10^6 matches) */
    uint32_t *rPos = sReduceAll + wave * (2 * kReduceCap);
    uint32_t *rId = rPos + kReduceCap;
    uint32_t rn = 0;                                /* staged pairs (wave-uniform) */
    int pendMatch[kWalkSets];
    uint32_t pendPos[kWalkSets];
#pragma unroll
    for (int s = 0; s < kWalkSets; s++) { pendMatch[s] = 0; pendPos[s] = 0; }

    auto report = [&](bool ended, const WLane &w, int s) {
        if (ended & (w.match != 0)) {
            if (REDUCE || kStagedPatch) {                  /* parked; stagePending() picks it up in uniform control flow */
                pendMatch[s] = w.match;
                pendPos[s] = w.pos;
            } else {
#if defined(PFAC_EXP_PATCH_LOCAL)       /* timing experiment: the patch stores land in one small window of the result vector; results are wrong */
                a.out[w.pos & 0xFFFFFu] = w.match;
#elif defined(PFAC_EXP_PATCH_NT)
                __builtin_nontemporal_store(w.match, &a.out[w.pos]);
#elif !defined(PFAC_EXP_NOSTORE)        /* timing experiment (tools/ab.py): walks without the patch store; results are wrong */
                a.out[w.pos] = w.match;
#endif
            }
        }
    };
    auto flushStaged = [&]() {                             /* all 64 lanes, uniform control flow */
        if (rn == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (REDUCE) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(a.reduceCount, rn);
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            for (uint32_t i = lane; i < rn; i += 64) {
                a.out[base + i] = (int)rId[i];
                a.reducePos[base + i] = (int)(a.reduceBase + rPos[i]);
            }
        } else {                                           /* full-result kernel: the staged matches overwrite their zeros, one store instruction for all of them */
            if ((uint32_t)lane < rn) a.out[rPos[lane]] = (int)rId[lane];
        }
        rn = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    auto stagePending = [&]() {
        if (!REDUCE && !kStagedPatch) return;
#pragma unroll
        for (int s = 0; s < kWalkSets; s++) {
            const bool has = pendMatch[s] != 0;
            const uint64_t m = __ballot(has);
            if (m) {
                const uint32_t cnt = (uint32_t)__popcll(m);
                if (rn + cnt > kReduceCap) flushStaged();
                if (cnt > kReduceCap) {                            /* pattern-dense input: this ballot alone is worth an atomic / a store instruction */
                    if (REDUCE) {
                        unsigned int base = 0;
                        if (lane == 0) base = atomicAdd(a.reduceCount, cnt);
                        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base) + laneRankIn(m);
                        if (has) { a.out[base] = pendMatch[s]; a.reducePos[base] = (int)(a.reduceBase + pendPos[s]); pendMatch[s] = 0; }
                    } else if (has) {
                        a.out[pendPos[s]] = pendMatch[s];
                        pendMatch[s] = 0;
                    }
                } else {
                    const uint32_t at = rn + laneRankIn(m);
                    if (has) { rPos[at] = pendPos[s]; rId[at] = (uint32_t)pendMatch[s]; pendMatch[s] = 0; }
                    rn = uni(rn + cnt);
                }
            }
        }
    };
    /* Does this wave expect LONG slots (pfac_context.h: wide buckets)?  Wave-uniform, decided from what its own walks meet:
     * off, a long slot whose header bytes match costs its walk one more trip (the unit is fetched then); once kSpecOnScore
     * walks have paid that, the units of wide buckets' slots are fetched with the headers and the window is kept 24 bytes
     * ahead -- until, 16 rounds in a row, fewer than a quarter of the units fetched were looked at (or hardly any was fetched).  Near-miss streams
     * (BASELINE config 5) run with it on from their first rounds; on text it stays off: a unit fetched for a walk that
     * dies on its edge byte is a gathered load of a cold line (Snort-style stream, always on: +2.8 % launch time). */
    constexpr bool kSpecKernel = WLane::kSpec;
    constexpr uint32_t kSpecOnScore = 16, kSpecOffRounds = 16;
#ifndef PFAC_SPEC_FORCE
#define PFAC_SPEC_FORCE -1                     /* measurement builds: 0 = never, 1 = always */
#endif
    /* the wave's vote on the walker of the handle's next launch: chunks during which eight or more of its walks went through long
     * slots or ran off their LDS bytes (near misses of long patterns: text has one such walk per chunk or so) minus the others */
    int advBalance = 0;
    uint32_t chunkEvents = 0;
    bool specOn = PFAC_SPEC_FORCE == 1 || (PFAC_SPEC_FORCE < 0 && kStageWalk && modeStage);
    uint32_t specScore = 0, specIdle = 0;
    auto walkIssue = [&]() {
#pragma unroll
        for (int s = 0; s < kWalkSets; s++) {
            if (alive[s]) walk[s].issue(wctx, specOn);
            (void)view;
            stLaneSteps += (uint32_t)__popcll(__ballot(alive[s]));
        }
        stRounds++;
#if PFAC_STATS
#pragma unroll
        for (int s = 0; s < kWalkSets; s++) {
            stSlotGathers += (uint32_t)__popcll(__ballot(alive[s] && walk[s].needSlot));
            if constexpr (!kStageWalk) stWinLoads += (uint32_t)__popcll(__ballot(alive[s] && walk[s].needWin));
        }
#endif
    };
    auto walkConsume = [&]() {
#pragma unroll
        for (int s = 0; s < kWalkSets; s++) {
            bool cont = false;
            if (alive[s]) {
                if constexpr (kStageWalk) cont = walk[s].consume(wctx);
                else cont = walk[s].consume(wctx, specOn);
            }
            if (kSpecKernel && PFAC_SPEC_FORCE < 0) {
                if (!specOn) {
                    /* on text nothing below ever happens: one test for all of it */
                    bool odd = walk[s].needExt;
                    if constexpr (kStageWalk) odd |= walk[s].ranOff;
                    if (__ballot(alive[s] & odd) != 0) {
                        chunkEvents += (uint32_t)__popcll(__ballot(alive[s] & odd));
                        if constexpr (kStageWalk) deepRecent += (uint32_t)__popcll(__ballot(alive[s] & walk[s].ranOff));
                        specScore += (uint32_t)__popcll(__ballot(alive[s] & walk[s].needExt));
                        if (specScore >= kSpecOnScore) { specOn = true; specScore = 0; specIdle = 0; }
                    }
                } else {
                    const uint32_t loaded = (uint32_t)__popcll(__ballot(alive[s] & walk[s].haveE)), used = (uint32_t)__popcll(__ballot(alive[s] & walk[s].usedE));
                    /* a round in which fewer than a quarter of the units fetched were looked at -- or hardly any was fetched at all: text -- */
                    chunkEvents += used;
                    specIdle = (loaded < 8u || used * 4u < loaded) ? specIdle + 1u : 0u;
                    if (specIdle >= kSpecOffRounds) { specOn = false; specIdle = 0; }
                }
            }
            if constexpr (!REDUCE && !kStageWalk && !kSpecKernel) {
                /* the window walker's evidence that its stream is full of near misses: walks through long slots, walks that outran their window */
                const bool odd = walk[s].tookLong | walk[s].needWin;
                if (__ballot(alive[s] & odd) != 0) chunkEvents += (uint32_t)__popcll(__ballot(alive[s] & odd));
            }
            report(alive[s] & !cont, walk[s], s);
            alive[s] = cont;
        }
        stagePending();
    };
    constexpr uint32_t kRefillBatch = (uint32_t)PFAC_REFILL_BATCH < kQCap / 2 ? (uint32_t)PFAC_REFILL_BATCH : kQCap / 2;   /* the ladder stops feeding a queue with less than 16 free entries */
    bool flushWalks = false;                        /* nothing left to filter: queued walks start however few they are */
#ifndef PFAC_WALK_GATE
#define PFAC_WALK_GATE 64                      /* 0: a walker round in every trip (rounds 2 - 5) */
#endif
#ifndef PFAC_WALK_GATE_TRIPS
#define PFAC_WALK_GATE_TRIPS 6
#endif
    constexpr bool kWalkGated = !REDUCE && !kStageWalk && kWalkSets == 1 && PFAC_WALK_GATE > 0;       /* (the compacted-output kernel's rounds are full: 116 of 128 lanes step per round on C3; gated, 0.589 -> 0.604 ms) */
    constexpr uint32_t kWalkGate = PFAC_WALK_GATE, kWalkGateTrips = PFAC_WALK_GATE_TRIPS;
    uint32_t walkIdleTrips = 0;
    /* hand verified queue entries to idle walker lanes */
    auto walkRefill = [&]() {
#pragma unroll
        for (int s = 0; s < kWalkSets; s++) {
            const uint64_t idle = __ballot(!alive[s]);
            /* text mode (kStageWalk): a queued candidate copies its bytes out of the stage when a lane takes it, and the next chunk is
             * not staged before that: the queue is handed out whenever the idle lanes can take all of it */
            const bool drainNow = kStageWalk && !modeStage && (uint32_t)__popcll(idle) >= qv - qh;
            /* stage mode: a buffer is overwritten when the last walk of its chunk has ended, so an entry should not wait for
             * sixteen idle lanes, and starting a walk there is a code and two LDS reads */
#ifndef PFAC_REFILL_MIN_STAGE
#define PFAC_REFILL_MIN_STAGE 6
#endif
            const uint32_t refillMin = (kStageWalk && modeStage) ? (uint32_t)PFAC_REFILL_MIN_STAGE : (uint32_t)PFAC_REFILL_MIN;
            if (((uint32_t)__popcll(idle) >= refillMin || drainNow) && qh != qv && (qv - qh >= kRefillBatch || flushWalks)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t rank = laneRankIn(idle);
                const bool take = !alive[s] & (rank < qv - qh);
                if (take) {
                    const uint32_t qi = (qh + rank) & kMask;
                    if constexpr (kStageWalk) {
#ifndef PFAC_LONG_JUMP
#define PFAC_LONG_JUMP 1
#endif
                        const bool longJump = PFAC_LONG_JUMP != 0 && specOn;
                        if (modeStage) walk[s].startStage(wctx, view, queue32[qi], HAS_SHORT ? sShort : nullptr, longJump);
                        else walk[s].startText(wctx, view, queue32[qi] & 0x7FFFFFFFu, view.addr[1] + ((uint32_t)s * 64u + (uint32_t)lane) * kWalkEntryBytes, HAS_SHORT ? sShort : nullptr, longJump);
                    } else {
                        const u32x2 eb = queueB[qi];
                        u32x4 ec = {0, 0, 0, 0};
                        if (kWideEntry) ec = queueC[qi];
                        walk[s].start(wctx, queue[qi], eb.x, eb.y, ec, HAS_SHORT ? sShort : nullptr);
                    }
                }
                alive[s] = alive[s] | take;
                const uint32_t idleLanes = (uint32_t)__popcll(idle);
                const uint32_t taken = idleLanes < qv - qh ? idleLanes : qv - qh;
                stStarts += taken;
                qh = uni(qh + taken);
            }
        }
    };
    auto anyAlive = [&]() {
        bool any = false;
#pragma unroll
        for (int s = 0; s < kWalkSets; s++) any |= alive[s];
        return __ballot(any) != 0;
    };

    /* Work is handed out dynamically and IN ORDER: block b serves part b % parts of the input and takes the
     * next piece of its part from a device counter (one per part, in a.work).  With PFAC_FRONT_LOG2 = -1 a part
     * is a contiguous 1/parts of the input: all waves of a part work inside a window of a few hundred KiB that
     * moves linearly through its part.  With G > 0 granules of 2^G pieces are dealt round-robin to the parts,
     * so the whole grid works inside ONE window that sweeps the input once.  Either way it is what the hardware
     * does for a grid of small blocks, worth ~10 % of HBM throughput over a static grid-stride assignment
     * (profiles/r01_stream_probe2_ordering.txt), and it balances the load.  (Workgroups are dealt round-robin
     * to the 8 XCDs; with two parts a counter is shared by the blocks of four XCDs.)
     * A piece is a span of kSpanChunks chunks claimed by a writer wave (kWriters > 0), or a single chunk claimed
     * by the scanning wave itself (kWriters == 0). */
    const uint32_t numChunks = n / kChunkBytes;
    const uint32_t numPieces = kWriters ? (numChunks + kSpanChunks - 1) >> kSpanLog2 : numChunks;
    /* parts: TWO for the full-result kernel, with granules of 16 spans: the narrowest front that two claim counters can
     * still serve (one counter saturates: ~90 atomics per microsecond; 512 writer waves ask 146 times per microsecond).
     * 16 parts x 4 spans (round 2) -> 2 x 16: C3 -6 %, C2 -9 %, and the buffer-placement classes disappear
     * (profiles/r03_experiments.md, section 4).  More for the compacted-output kernel, whose waves claim granules of
     * chunks themselves and would queue up at the counters */
    constexpr uint32_t kParts = REDUCE ? (uint32_t)PFAC_REDUCE_PARTS : (uint32_t)pfac::kWorkParts;
    const uint32_t parts = gridDim.x < kParts ? gridDim.x : kParts;
    const uint32_t part = blockIdx.x % parts;
    constexpr bool kFrontOn = PFAC_FRONT_LOG2 >= 0;
    constexpr uint32_t kFront = kFrontOn ? PFAC_FRONT_LOG2 : 0;
    const uint32_t partBegin = kFrontOn ? 0u : (uint32_t)((uint64_t)numPieces * part / parts);
    const uint32_t pieceEnd = kFrontOn ? numPieces : (uint32_t)((uint64_t)numPieces * (part + 1) / parts);
    auto pieceOf = [&](uint32_t v) {                       /* v-th piece of this block's part */
        if (kFrontOn) return ((((v >> kFront) * parts + part) << kFront) | (v & ((1u << kFront) - 1u)));
        return partBegin + v;
    };
    unsigned int *const counter = a.work + part * 32;

    if (kWriters && wave >= kScanners) {
        /* ---- writer wave: claim, zero-fill, publish.  One span in flight per writer: keeping two in flight (the
         * next span claimed and issued before the previous one is waited for) was worth 2..4 % while the scanning
         * waves stalled on their own loads, is worth nothing since they do not, and costs 4 % when the launch is bound
         * by the result stream (profiles/r02_ab_prefetch_registers_and_writers.txt, r02_ab_list_refill_order.txt).
         * A writer spends a third of its time on the claim (PFAC_TIMING build: 27 % in the device atomic, 9 % waiting for
         * the other writer's), but asking for the next span while the zeros of this one drain only moves that time into
         * the store queue: the zeros then take that much longer to issue, the launch takes the same time
         * (profiles/r03_experiments.md) -- the result stream is bound by the memory system, not by the writers. */
        const i32x4 zero = {0, 0, 0, 0};
#if PFAC_TIMING     /* profile build: where a writer wave's time goes */
        uint32_t wt[6] = {0, 0, 0, 0, 0, 0};
        uint64_t wLast = __builtin_readcyclecounter();
#define PFAC_WTICK(k) do { __builtin_amdgcn_sched_barrier(0); const uint64_t tNow = __builtin_readcyclecounter(); wt[k] += (uint32_t)(tNow - wLast); wLast = tNow; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PFAC_WTICK(k) do { } while (0)
#endif
        for (uint32_t k = (uint32_t)(wave - kScanners);; k += kWriters) {
#if PFAC_TIMING
            while (ldsLoad(&ctl->claimTurn) != k) __builtin_amdgcn_s_sleep(8);
            PFAC_WTICK(0);
#endif
            for (;;) {                                      /* my turn to claim, and a ring slot nobody still reads */
                const uint32_t turn = ldsLoad(&ctl->claimTurn), pops = ldsLoad(&ctl->popCount);
                if (turn == k && k < (pops >> kSpanLog2) + kRunAhead) break;
                __builtin_amdgcn_s_sleep(8);
            }
            PFAC_WTICK(1);
            unsigned int v = 0;
            if (lane == 0) v = atomicAdd(counter, 1u);
            const uint32_t span = pieceOf(uni(v));
            PFAC_WTICK(2);
            ldsStore(&ctl->claimTurn, k + 1);
            if (span >= pieceEnd) {
                if (lane == 0) atomicMin(&ctl->endSpan, k);
                break;
            }
            const uint32_t c0 = span << kSpanLog2;
            const uint32_t cN = c0 + kSpanChunks < numChunks ? c0 + kSpanChunks : numChunks;
            i32x4 *o4 = reinterpret_cast<i32x4 *>(a.out + (size_t)c0 * kChunkBytes);
            const uint32_t stores = (cN - c0) * (kChunkBytes * 4 / 1024);                 /* 1 KiB per instruction */
            /* non-temporal: plain stores run the launch 10 % slower, stores with a wider scope (sc0 / sc1) 2-3 times
             * (profiles/r02_ab_zero_store_policy.txt) */
            for (uint32_t i = 0; i < stores; i++) __builtin_nontemporal_store(zero, &o4[i * 64 + lane]);
            PFAC_WTICK(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                              /* the zeros are in L2 */
            PFAC_WTICK(4);
            while (ldsLoad(&ctl->pubCount) != k) __builtin_amdgcn_s_sleep(2);             /* publish in order */
            ldsStore(&ctl->ring[k & (kRing - 1)], span);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            ldsStore(&ctl->pubCount, k + 1);
            PFAC_WTICK(5);
        }
#if PFAC_TIMING
        if (lane == 0)
            for (int k = 0; k < 6; k++) atomicAdd(reinterpret_cast<unsigned long long *>(a.work + pfac::kStatsWord) + 20 + k, (unsigned long long)wt[k]);
#endif
    } else {
    /* ---- scanning wave */
    /* the ends of the input first (ScanArgs::endsIn), 64 positions per wave, dealt to the first scanning wave of every
     * block, then the second, ...: such a wave joins the scan a few microseconds late, which the others make up for --
     * chunks are claimed, not assigned */
    if (a.endsIn != nullptr) {
        const uint32_t lenA = a.endsA1 - a.endsA0, total = lenA + (a.endsB1 - a.endsB0);
        const WCtx ends(a);
        for (uint32_t first = ((uint32_t)wave * gridDim.x + blockIdx.x) * 64u; first < total; first += gridDim.x * (uint32_t)kScanners * 64u) {
            const uint32_t i = first + (uint32_t)lane;
            if (i < total) {
                const uint32_t p = i < lenA ? a.endsA0 + i : a.endsB0 + (i - lenA);
                const int m = boundedWalk<TEX>(ends, a.endsIn, p, a.endsReadable);
                if (!REDUCE) {
                    a.endsOut[p] = m;
                } else if (m > 0) {
                    const unsigned int at = atomicAdd(a.reduceCount, 1u);
                    a.out[at] = m;
                    a.reducePos[at] = (int)p;        /* endsIn is the caller's first byte */
                }
            }
        }
    }
    /* ticket for the next chunk (lane 0 holds the answer): cheap, asked for one chunk ahead ... */
    /* Without writer waves the tickets come from the part's device counter, a granule of the front (adjacent chunks)
     * per atomic: the wave waits for the atomic's answer -- and, the counter being in-order, for its own loads in
     * flight -- so it asks once per granule, not once per chunk. */
    constexpr uint32_t kTicketBatch = kWriters ? 1u : (kFrontOn ? (1u << kFront) : 1u);
    uint32_t ticketNext = 0, ticketEnd = 0;
    auto pop = [&]() -> unsigned int {
        unsigned int v = 0;
        if (kWriters) {
            if (lane == 0) v = atomicAdd(&ctl->popCount, 1u);
            return v;
        }
        if (ticketNext == ticketEnd) {
            if (lane == 0) v = atomicAdd(counter, kTicketBatch);
            ticketNext = uni(v);
            ticketEnd = ticketNext + kTicketBatch;
        }
        return ticketNext++;
    };
    /* ... and turned into a chunk number when its data is to be prefetched: waits for the writers if they are
     * behind (then the launch is bound by the result stream, as it should be).  kEnd = the part is finished. */
    auto resolve = [&](uint32_t ticket) -> uint32_t {
        if (!kWriters) {
            const uint32_t c = pieceOf(ticket);
            return c < pieceEnd ? c : kEnd;
        }
        const uint32_t slot = ticket >> kSpanLog2;
        for (;;) {
            if (slot < ldsLoad(&ctl->pubCount)) break;
            if (slot >= ldsLoad(&ctl->endSpan)) return kEnd;
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t c = (uni(ldsLoad(&ctl->ring[slot & (kRing - 1)])) << kSpanLog2) | (ticket & (kSpanChunks - 1u));
        return c < numChunks ? c : kEnd;                   /* the last span of the input may be partial */
    };
    /* The chunk in flight lives in nine vector registers that the COMPILER DOES NOT KNOW ABOUT (v119..v127: the
     * kernel is compiled for fewer registers, kCompilerVgprs): two tiles, 1 KiB per load instruction, and the 32
     * bytes (48 of them are used) behind the chunk, one dword in each of the lanes 0..15.  Left to the register allocator they were
     * copied between two register sets on every trip of the loop that did not stage a chunk, and a copy of the
     * destination of a load in flight waits for it -- and, the wait counter being in-order, for the walkers'
     * loads just issued: a third of a scanning wave's time (PFAC_TIMING build).  Issued and read through inline
     * assembly, they are waited for in one place: the top of the scan loop. */
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"        /* "clobber list contains reserved registers": that is the point */
    auto prefetchChunk = [&](uint32_t c) {
        /* the lane's offsets are computed on the spot (volatile: loop invariants the compiler would keep in -- or spill
         * from -- registers that the walkers' state needs) */
        uint32_t off, offHalo;
        asm volatile("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(off) : "v"(lane), "s"(c * (uint32_t)kChunkBytes));
        /* the bytes behind the chunk: 64 of them (lanes 0..15: 48 are staged) -- in stage mode kWalkHalo, a dword in each of the lanes 0..31 */
        if constexpr (kStageWalk) asm volatile("v_and_b32 %0, %3, %1\n\tv_lshl_or_b32 %0, %0, 2, %2" : "=&v"(offHalo) : "v"(lane), "s"((c + 1u) * (uint32_t)kChunkBytes), "s"(modeStage ? 31u : 15u));
        else asm volatile("v_and_b32 %0, 15, %1\n\tv_lshl_or_b32 %0, %0, 2, %2" : "=&v"(offHalo) : "v"(lane), "s"((c + 1u) * (uint32_t)kChunkBytes));
        static_assert(kTilesPerIter == 2, "two tile registers are reserved");
#ifndef PFAC_INPUT_POLICY
#define PFAC_INPUT_POLICY ""                   /* cache policy of the chunk loads (" nt", " sc1", ...): measurement builds */
#endif
        asm volatile("global_load_dwordx4 v[120:123], %0, %2" PFAC_INPUT_POLICY "\n\t"
                     "global_load_dwordx4 v[124:127], %0, %2 offset:1024" PFAC_INPUT_POLICY "\n\t"
                     "global_load_dword v119, %1, %2"
                     :: "v"(off), "v"(offHalo), "s"(a.in)
                     : "memory", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    };
#pragma clang diagnostic pop
    /* the prefetched chunk -> ordinary registers, a tile at a time (the copies only live while level 1 runs) */
    auto takeTile0 = [&](u32x4 &d0, uint32_t &d1x) {
        asm volatile("v_mov_b32 %0, v120\n\tv_mov_b32 %1, v121\n\tv_mov_b32 %2, v122\n\tv_mov_b32 %3, v123\n\tv_mov_b32 %4, v124"
                     : "=v"(d0.x), "=v"(d0.y), "=v"(d0.z), "=v"(d0.w), "=v"(d1x) :: "memory");
    };
    auto takeTile1 = [&](u32x4 &d1, uint32_t &halo) {
        asm volatile("v_mov_b32 %0, v124\n\tv_mov_b32 %1, v125\n\tv_mov_b32 %2, v126\n\tv_mov_b32 %3, v127\n\tv_mov_b32 %4, v119"
                     : "=v"(d1.x), "=v"(d1.y), "=v"(d1.z), "=v"(d1.w), "=v"(halo) :: "memory");
    };

#if PFAC_ABLATE == 1
    uint32_t ablateSink = 0;
#endif
    uint32_t chunk = resolve(uni(pop()));
    uint32_t nextTicket = uni(pop());
    if (chunk != kEnd) prefetchChunk(chunk);

    /* The staged chunk: level-1 hits not yet listed (per lane), listed codes not yet tested [listAt, listEnd),
     * and its position in the input.  One loop, one copy of every stage: each trip starts with a walker round;
     * a new chunk is staged only when the previous one is completely listed and tested, and list entries are
     * tested only while the walk queue has room for a full pass -- otherwise the trip just walks. */
    uint32_t hits = 0;                          /* bit 16 * tt + i: position i of this lane in tile tt of the staged chunk */
    /* the two constants of the level-1 test in VECTOR registers: an instruction with a scalar or literal operand
     * issues at ~0.6 of the rate of the same instruction on vector registers (tools/valu_probe2.hip: v_lshrrev
     * 1.10 vs 1.78 ns, v_mul_u32_u24 1.76 vs 2.03 ns per wave and SIMD), and these two run 2048 times per chunk */
    uint32_t vShift3, vGram3Mul;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vShift3) : "s"(REDUCE ? 0xFFFCu : lds.shift3));      /* the address mask of the level-1 bitmap: (dwords - 1) << 2 (gram1: 64 KiB) */
    asm volatile("v_mov_b32 %0, %1" : "=v"(vGram3Mul) : "s"(REDUCE ? pfac::kGram1Mul : pfac::kGram3Mul));
    uint32_t listAt = 0, listEnd = 0, stagedBase = 0;
    bool freshChunk = false;                    /* level 1 of the staged chunk has just run: `hits` holds all of its hits */
    uint32_t *sDense = sDenseAll + wave * kDenseStage;     /* pattern-dense chunks of this wave, not yet on the launch's list */
    uint32_t nDense = 0;
    auto flushDense = [&]() {                   /* wave-uniform control flow */
        if (nDense == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(a.work + a.denseWord, nDense);
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        if ((uint32_t)lane < nDense) {
            uint32_t at;                                   /* computed on the spot: not an address the compiler keeps (or spills) across the scan loop */
            asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(at) : "v"(lane), "s"((uint32_t)(reinterpret_cast<unsigned char *>(sDense) - smem)));
            a.denseList[base + lane] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(at);
        }
        nDense = 0;
    };
    uint32_t ladderIdle = 0, ladderSkip = 0;    /* wave-uniform: batches in a row that the ladder did not thin out / batches left to walk untested */
    /* this lane's code of the list round that starts at entry `first` (wave-uniform); the address is computed on the spot */
    const uint32_t listBaseBytes = (uint32_t)(reinterpret_cast<unsigned char *>(list) - smem);
    auto listCode = [&](uint32_t first) -> uint32_t {
        uint32_t addr;
        asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(addr) : "v"(lane), "s"(listBaseBytes + 2u * first));
        return (uint32_t)*reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>(addr);
    };
#if PFAC_TIMING     /* profile build: shader-clock cycles this wave spends in each stage (s_memtime at the stage boundaries) */
    uint32_t tm[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tLast = __builtin_readcyclecounter();
#define PFAC_TICK(k) do { __builtin_amdgcn_sched_barrier(0); const uint64_t tNow = __builtin_readcyclecounter(); tm[k] += (uint32_t)(tNow - tLast); tLast = tNow; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PFAC_TICK(k) do { } while (0)
#endif
    /* the tail hash of a candidate (pfac::tailRoll): `start` rolled over the nb (a multiple of four, <= kTailMaxBytes) staged bytes from offset `from` of the chunk --
     * nine dwords of the stage in one go, then eight steps in registers */
    auto tailRun = [&](uint32_t start, bool chk, uint32_t from, uint32_t nb) -> uint32_t {
        static_assert(pfac::kTailMaxBytes % 4 == 0 && pfac::kTailMaxBytes >= 8 && pfac::kTailMaxBytes <= 32, "whole dwords, four bytes a step; three bits of the device-memory entry say how many");
        constexpr int kSteps = pfac::kTailMaxBytes / 4;
        const uint32_t b = chk ? from : 0u, at4 = b >> 2, shb = b & 3u;
        uint32_t w[kSteps + 1], run = start;
#pragma unroll
        for (int k = 0; k < kSteps + 1; k++) w[k] = stage[at4 + (uint32_t)k];
#pragma unroll
        for (int k = 0; k < kSteps; k++) {
            const uint32_t x = __builtin_amdgcn_alignbyte(w[k + 1], w[k], shb);
            const uint32_t r1 = (run ^ x) * pfac::kLadMul;
            run = (chk && 4u * (uint32_t)k < nb) ? r1 : run;
        }
        return run;
    };
    /* VETO = 2: the stopped candidates of one ladder batch whose buckets of the device-memory tail table are on their way (wave-uniform flag; per lane the
     * bucket, the candidate's offset in the staged chunk | 1 << 31, the ladder hash that stopped it) */
    u32x4 pendE = {0, 0, 0, 0};
    uint32_t pendCode = 0, pendHash = 0;
    bool pendAny = false;
    uint32_t pendCount = 0;                        /* ... how many they are: queue entries a later batch of the same trip must leave them (wave-uniform) */
    /* ... looked at at the top of the next trip: the chunk is still staged (the next one is staged further down the trip), the queue still has the room the
     * batch was cut to.  A candidate whose bytes hash like the rest of its pattern -- or that has no entry, or whose bytes are not all staged -- walks. */
    auto tailResolve = [&]() {
        const bool mine = (pendCode >> 31) != 0u;
        const uint32_t o = pendCode & 0x7FFFFFFFu;
        const bool m1 = pendE.x == pendHash && (pendE.y & pfac::kTailGFromMask) != 0u, m2 = pendE.z == pendHash && (pendE.w & pfac::kTailGFromMask) != 0u;
        const uint32_t want = m1 ? pendE.y : pendE.w;
        const uint32_t nb = ((want & 7u) + 1u) * 4u, from = o + ((want >> 3) & 0xFFu);
        const bool chk = mine && (m1 | m2) && from + nb + 4u <= (uint32_t)kChunkBytes + 4u * (uint32_t)kHaloDwords;
        const bool vetoed = chk && ((tailRun(pendHash, chk, from, nb) ^ want) & ~pfac::kTailGInfoMask) != 0u;
        /* a vetoed candidate is a near miss of a long pattern: this kernel's evidence of what its stream is (the walks it spares are what the other
         * kernels count: the vote at the end of a chunk) */
        chunkEvents += (uint32_t)__popcll(__ballot(vetoed));
        const bool keep = mine && !vetoed;
        const uint64_t keepMask = __ballot(keep);
        if (keepMask != 0) {
            if (keep) {
                const uint32_t at = o >> 2, sh = o & 3u;
                uint32_t e[6];
#pragma unroll
                for (int k = 0; k < 6; k++) e[k] = stage[at + (uint32_t)k];
                const uint32_t qi = (qv + laneRankIn(keepMask)) & kMask;
                const u32x4 entry = {stagedBase + o, __builtin_amdgcn_alignbyte(e[1], e[0], sh), __builtin_amdgcn_alignbyte(e[2], e[1], sh), __builtin_amdgcn_alignbyte(e[3], e[2], sh)};
                const u32x2 entryB = {__builtin_amdgcn_alignbyte(e[4], e[3], sh), __builtin_amdgcn_alignbyte(e[5], e[4], sh)};
                queue[qi] = entry;                                  /* (20-byte entries: kEntry) */
                queueB[qi] = entryB;
            }
            qv = uni(qv + (uint32_t)__popcll(keepMask));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        pendAny = false;
        pendCode = 0u;
        pendCount = 0u;
    };
    /* One loop, one copy of every stage.  A trip: (1) finish the walkers' transitions issued one trip ago; (2) refill
     * idle walker lanes and issue the next transitions; (3) if the staged chunk is completely listed and tested: level
     * 1 over the next chunk (prefetched one chunk ago), then the prefetch of the one after it; (4) the next <= kListCap
     * hits -> list; (5) passes while the walk queue has room.  The order matters.  The wait counter for vector memory
     * is in-order, so the one wait at the top of a trip is for everything issued in the trip before; the walkers'
     * loads are what it mostly waits for (gathered, from L2 or beyond), so they are issued first and have (3)-(5) to
     * land: issued after level 1 they cost C3 1.5 % and C5 4 % (profiles/r02_ab_list_refill_order.txt). */
    for (;;) {
        PFAC_TICK(7);
        /* every load of the previous trip: the walkers' slots, which (1) is about to use, and -- older -- the
         * prefetched chunk.  Written out (the compiler's own waits sit inside `if (alive)` blocks and it would
         * add more further down for loads it cannot prove finished), and it is all this loop ever waits for:
         * s_waitcnt vmcnt(0), expcnt and lgkmcnt untouched */
        __builtin_amdgcn_s_waitcnt(0x0F70);
        PFAC_TICK(8);
        if constexpr (kVetoG) { if (pendAny) tailResolve(); }
        flushWalks = chunk == kEnd && listAt == listEnd;
        /* The walkers of the window-walker kernels do not run in every trip (round 6).  A round -- consume, refill, issue -- costs the wave the
         * same ~250 instructions for five live walks as for sixty-four, and behind the prefilter a chunk leaves six to eight walks (text: 3.3 M
         * walks per GiB, 22 lane steps per chunk: rounds were a third full; the near-miss streams behind the veto the same), on streams whose
         * launches are bound by instruction issue.  A round runs when live + queued walks would fill kWalkGate lanes, when the queue is half
         * full, at the end of the wave's input -- and after kWalkGateTrips trips without one, so that no walk waits long.  The slots loaded by
         * the last round stay in their registers meanwhile (the top-of-trip wait has seen them arrive). */
        bool walkRound = true;                                  /* wave-uniform */
        if constexpr (kWalkGated) {
            const uint32_t live = (uint32_t)__popcll(__ballot(alive[0])), queued = qv - qh;
            walkRound = flushWalks || live + queued >= kWalkGate || queued >= kQCap / 2 || walkIdleTrips >= kWalkGateTrips;
            walkIdleTrips = walkRound ? 0u : walkIdleTrips + 1u;
        }
        if (walkRound) {
            walkConsume();
            PFAC_TICK(0);
            /* ---- 2. hand idle walker lanes new positions, start the next transition of
             *         every live walk */
            walkRefill();
            PFAC_TICK(1);
            walkIssue();
        }
        PFAC_TICK(2);
        if (listAt == listEnd && __ballot(hits != 0) == 0) {
            /* kStageWalk: the next chunk goes into the buffer of the chunk before the one just filtered -- once no walk reads that
             * one any more (none of its entries still queued, none of its walks alive); until then the trip just walks */
            bool stageFree = true;
            if constexpr (kStageWalk) {
                /* which mode the stream asks for: stage mode while the wave meets long slots (specOn) or its walks run off their
                 * entries, and for a while after; the region changes hands when nothing is queued and nothing walks */
#ifndef PFAC_FORCE_MODE
#define PFAC_FORCE_MODE -1                     /* measurement builds: 0 = text mode only, 1 = stage mode only */
#endif
                const bool wantStage = PFAC_FORCE_MODE >= 0 ? PFAC_FORCE_MODE == 1 : (specOn | (deepRecent >= 32u) | (stageHold != 0));
                if (wantStage != modeStage) {
                    if (qh == qv && !anyAlive()) {
                        modeStage = wantStage;
                        cur = 0;
                        stage = stage0;
                        qEnd[0] = qEnd[1] = qv;
                        if (modeStage) {
                            stageHold = 8;
                            /* the chunk in flight was fetched with text mode's 64 bytes behind it: fetch it again with kWalkHalo, stage
                             * it in the next trip (behind the loop's wait) */
                            if (chunk != kEnd) { prefetchChunk(chunk); stageFree = false; }
                        }
                    } else {
                        stageFree = false;                         /* no new chunk until the walks of the old mode are through */
                    }
                }
                if (!modeStage) {
                    stageFree = stageFree && qh == qv;             /* text mode: what is queued has not copied its bytes out of the stage yet */
                } else if (stageFree) {
                    const uint32_t qe = cur ? qEnd[0] : qEnd[1];
                    bool reads = false;
#pragma unroll
                    for (int s = 0; s < kWalkSets; s++) reads |= alive[s] & (walk[s].inB == (cur == 0));
                    stageFree = (int)(qe - qh) <= 0 && __ballot(reads) == 0;
                }
            }
#ifdef PFAC_COUNT_STALLS                       /* measurement build: the level-1 statistic counts the trips that could not stage the next chunk (its buffer still read) */
            if (chunk != kEnd && !stageFree) stHits++;
#endif
            if (chunk == kEnd) {
                if (qh == qv && !anyAlive()) break;      /* nothing staged, queued or walking */
            } else if (stageFree) {
                /* ---- 3. next chunk: ask for the chunk after next; without writer waves: zero stores, 16 B per
                 *         lane, 1 KiB contiguous per instruction (older than every load of a walk that starts in
                 *         this chunk) */
                const unsigned int afterNext = pop();
                if (!REDUCE) { advBalance += chunkEvents >= 8u ? 1 : -1; chunkEvents = 0; }
                if constexpr (kStageWalk) {
                    if (modeStage) {
                        if (cur) qEnd[1] = qv; else qEnd[0] = qv;  /* whatever the chunk just filtered put on the queue lies in front of qv */
                        cur ^= 1u;
                        stage = stage0 + cur * (uint32_t)kStageWordsK;
                        if (!specOn && deepRecent < 32u && stageHold != 0) stageHold--;
                    }
                    if ((deepRecent | specScore) != 0) {            /* per chunk: a quarter of what is left */
                        deepRecent -= (deepRecent + 3u) >> 2;
                        specScore -= (specScore + 3u) >> 2;
                    }
                }
                if (!REDUCE && !kWriters) {
                    i32x4 *o4 = reinterpret_cast<i32x4 *>(a.out + (size_t)chunk * kChunkBytes);
                    const i32x4 zero = {0, 0, 0, 0};
#pragma unroll
                    for (int k = 0; k < 4 * kTilesPerIter; k++) __builtin_nontemporal_store(zero, &o4[k * 64 + lane]);
                }
                /* filter level 1: lane l owns bytes 16l..16l+15 of each tile, one LDS bit test per position.
                 * The chunk also goes to LDS (lane l -> bytes 16l.. of its tile, lanes 0..11 also the 48 bytes behind
                 * it): an entry of the walk queue needs 20 bytes from an arbitrary offset. */
#pragma unroll
                for (int tt = 0; tt < kTilesPerIter; tt++) {
                    u32x4 dt;
                    uint32_t follow;                   /* lane 0: the dword behind this tile */
                    if (tt == 0) takeTile0(dt, follow);
                    else takeTile1(dt, follow);
                    {   /* the lane's place in the stage, computed on the spot (see the halo below) */
                        uint32_t at16;
                        const uint32_t tileBase = (uint32_t)(reinterpret_cast<unsigned char *>(stage) - smem) + (uint32_t)tt * (uint32_t)kTileBytes;
                        asm volatile("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(at16) : "v"(lane), "s"(tileBase));
                        *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(at16) = dt;
                    }
                    if (tt == kTilesPerIter - 1 && lane < (kStageWalk ? (modeStage ? kHaloDwords : 16) : kHaloDwords)) {
                        /* the address is computed on the spot (volatile: not hoisted out of the loop into a register
                         * that lives -- or is spilled -- across it) */
                        uint32_t at;
                        const uint32_t haloBase = (uint32_t)(reinterpret_cast<unsigned char *>(stage + kTilesPerIter * 256) - smem);   /* smem is LDS address 0 */
                        asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(at) : "v"(lane), "s"(haloBase));
                        *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(at) = follow;
                    }
                    const uint32_t dw[4] = {dt.x, dt.y, dt.z, dt.w};
                    /* the first dword of the next lane: one DPP move (wave_shl:1), no lane-number register for a bpermute */
                    uint32_t nxtLane = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dw[0], 0x130, 0xf, 0xf, false);
                    const uint32_t wrap = (uint32_t)__builtin_amdgcn_readfirstlane((int)follow);
                    if (lane == 63) nxtLane = wrap;
                    /* kBatch positions at a time (that many LDS reads in flight); the scheduling barrier keeps the
                     * batches apart, or their temporaries pile up past the register budget.  Per position: the gram
                     * (a shift or v_alignbyte), v_mul_u32_u24, shift + AND = dword address, ds_read_b32, a shift by the
                     * gram (mod 32: the bit), v_alignbit to push the bit into the mask.  Plain VOP2 instructions
                     * wherever possible: they issue twice as fast as VOP3 ones here (tools/valu_probe.hip). */
#ifndef PFAC_L1_BATCH
#define PFAC_L1_BATCH 8                        /* ... of the full-result kernel, whose walkers (StageLane) leave it the registers for more */
#endif
                    constexpr int kBatch = kStageWalk ? PFAC_L1_BATCH : 8;
#pragma unroll
                    for (int b0 = 0; b0 < 16; b0 += kBatch) {
                        uint32_t word[kBatch], xs[kBatch + 1];
#pragma unroll
                        for (int q = 0; q < kBatch; q++) {
                            const int j = (b0 + q) >> 2, i = (b0 + q) & 3;
                            const uint32_t nx = j < 3 ? dw[(j + 1) & 3] : nxtLane;
                            /* bytes pos..pos+2 in the low 24 bits (the multiply ignores the rest) */
                            const uint32_t x = i == 0 ? dw[j] : i == 1 ? dw[j] >> 8 : __builtin_amdgcn_alignbyte(nx, dw[j], i);
                            /* dword of the 3-gram: the top bits of the 24 x 24 -> 32 bit product, as a byte address.  (The high half of
                             * the 48-bit product -- one v_mul_hi_u32_u24 and an AND -- would save an instruction, but the first byte of
                             * the gram hardly reaches it: level-1 hits went from 5 % to 18 % of the text stream.) */
                            const uint32_t product = (uint32_t)__umul24(x, vGram3Mul);   /* __umul24 returns int: shifts must be logical */
                            /* byte address of the dword = bits 18.. of the product, times four = the product's high half AND (dwords - 1) << 2 --
                             * one SDWA instruction (a shift and an AND otherwise: round 5's full-result kernels) */
                            uint32_t addr;
                            asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(addr) : "v"(product), "v"(vShift3));
                            word[q] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(addr + (REDUCE ? kGram1LdsOffset : 0u));
                            xs[q] = x;
                        }
                        /* the second bit of a 3-gram is numbered by the low five bits of its SECOND byte: the first byte of the
                         * next position -- whose gram (or whose raw dword) is at hand, no shift needed */
                        xs[kBatch] = b0 + kBatch < 16 ? dw[(b0 + kBatch) >> 2] : nxtLane;
#pragma unroll
                        for (int q = 0; q < kBatch; q++) {
                            if (REDUCE) hits = __builtin_amdgcn_alignbit(word[q] >> (xs[q] & 31u), hits, 1);    /* one bit per 3-gram */
                            else hits = __builtin_amdgcn_alignbit((word[q] >> (xs[q] & 31u)) & (word[q] >> (xs[q + 1] & 31u)), hits, 1);   /* both bits set: bit 0 enters at the top */
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                stagedBase = chunk * kChunkBytes;
                if constexpr (kStageWalk) { if (cur) view.base[1] = stagedBase; else view.base[0] = stagedBase; }
                freshChunk = true;
                PFAC_TICK(3);
                /* the chunk registers are free again: prefetch the next chunk.  Past the end the last chunk is
                 * loaded again, not nothing (it is never taken) */
                const uint32_t next = resolve(nextTicket);
                PFAC_TICK(6);
                prefetchChunk(next != kEnd ? next : chunk);
                chunk = next;
                nextTicket = uni(afterNext);
#if PFAC_ABLATE == 1
                ablateSink |= hits;
                hits = 0;
#endif
            }
        }
        /* ---- 4. the lanes' hits -> one list of 16-bit codes (lane << 5 | bit), slot = prefix sum of the hit
         *         counts; hits beyond the list's capacity stay in `hits` for the next trip */
        /* (full-result kernel: also when fewer than kMergeMin tested candidates are left over while the chunk still has hits to list -- the next hits are
         * listed behind them and one batch of the ladder takes them together.  A batch costs the wave the same ~600 instructions for 20 candidates as
         * for 64, and a chunk with more level-1 hits than the list holds -- a Snort-scale set over alphanumeric text: 170 -- used to pay for two or
         * three of them: BASELINE config 5's stream over the 31 000-pattern set, 2.3 batches per chunk -> 1.) */
        if ((listAt == listEnd || (!REDUCE && listEnd - listAt < kMergeMin)) && __ballot(hits != 0) != 0) {
            const uint32_t carry = listEnd - listAt;                /* tested candidates that stay in front of the new hits (wave-uniform) */
            if (carry != 0 && listAt != 0) {                        /* ... at the list's head: carry < kMergeMin <= 64, one code per lane */
                const uint32_t code = (uint32_t)lane < carry ? listCode(listAt) : 0u;
                if ((uint32_t)lane < carry) list[lane] = (uint16_t)code;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            const uint32_t cnt = (uint32_t)__builtin_popcount(hits);
            const uint32_t incl = waveInclusiveScan(cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t idx = carry + incl - cnt;
            PFAC_TICK(9);
            const bool dense = !REDUCE && freshChunk && total > kDenseHits && a.denseList != nullptr;      /* wave-uniform */
            if (dense) {
                /* a pattern-dense chunk (most positions pass level 1: patterns of one or two bytes over text, a run of
                 * one byte that is a pattern): listing, testing and queueing every position costs more than walking them
                 * all.  The chunk goes on the launch's dense list and the tiled kernel behind this one does it. */
                if (lane == 0) sDense[nDense] = stagedBase / (uint32_t)kChunkBytes;
                nDense++;
                if (nDense == kDenseStage) flushDense();
                hits = 0;
            }
            freshChunk = false;
            while (hits != 0 && idx < kListCapK) {       /* divergent: as many rounds as the busiest lane has hits */
                list[idx] = (uint16_t)(((uint32_t)lane << 5) | (uint32_t)__builtin_ctz(hits));
                idx++;
                hits &= hits - 1;
            }
            PFAC_TICK(10);
            const uint32_t listed = dense ? 0u : (total < kListCapK - carry ? total : kListCapK - carry);
#ifndef PFAC_COUNT_STALLS
            stHits += listed;
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            PFAC_TICK(4);
            /* ---- 5a. every listed hit: its first four bytes against level 4 of the prefix ladder (are they a pattern
             * prefix at all?), the length-3 bitmap and the exact 2-byte bitmap.  Survivors stay in the list, compacted
             * in place (a wave's LDS accesses execute in order: every lane has read its code before any lane writes, and
             * the k-th round writes below the codes it has read); bit 15 = "walk, whatever follows" (an S node at depth
             * 4, or a pattern of up to three bytes matches here). */
            uint32_t kept = carry;
            for (uint32_t base = 0; base < listed; base += 64u) {
                const bool act = base + (uint32_t)lane < listed;
                const uint32_t code = act ? listCode(carry + base) : 0u;
                const uint32_t o = ((code & 0x10u) << 6) | ((code >> 1) & 0x3F0u) | (code & 0xFu);       /* byte offset inside the chunk: tile, lane, position */
                const uint32_t at = o >> 2, sh = o & 3u;
                const uint32_t x = __builtin_amdgcn_alignbyte(stage[at + 1], stage[at], sh);
                const uint32_t h = REDUCE ? x * pfac::kLadMul0 : (x * pfac::kLadMul0) ^ a.ladderSalt;      /* pfac::ladderStart; prefix4 is not salted */
                uint32_t sHit, gHit;
                if (REDUCE) {                                      /* every 4-byte pattern prefix walks: prefix4, two probes (LDS address 0) */
                    auto probe4 = [&](uint32_t v) -> uint32_t {
                        const uint32_t idx = v >> (32 - pfac::kPrefix4Log2);
                        return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((idx >> 3) & ~3u) >> (idx & 31u);
                    };
                    sHit = probe4(h) & probe4(h * pfac::kLadMulS) & 1u;
                    gHit = 0;
                } else {
                    const uint32_t w = ladWord(h);
                    sHit = ladStop(w, h) & 1u;
                    gHit = ladGoOn(w, h) & ladGoOn2(w, h) & 1u;               /* depth 4: G nodes set two bits */
                }
                uint32_t decided = sHit | (testBit(sFinal3, (uint32_t)__umul24(x, pfac::kFinal3Mul) >> lds.shiftF3) &
                                           testBit(sFinal3, (uint32_t)__umul24(x, pfac::kFinal3Mul2) >> lds.shiftF3));
                if (HAS_SHORT) decided |= testBit(sShort, x & 0xFFFFu);
                const bool keep = act && (decided | gHit) != 0;
                const uint64_t keepMask = __ballot(keep);
                if (keep) list[kept + laneRankIn(keepMask)] = (uint16_t)(code | (decided << 15));
                kept = uni(kept + (uint32_t)__popcll(keepMask));
            }
            listAt = 0;
            listEnd = kept;
            stCand += kept - carry;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            PFAC_TICK(11);
        }
        /* ---- 5b. the prefix ladder, one candidate per lane: cut its 20 bytes out of the stage and test the prefixes
         * of 6, 8, ..., 20 bytes against the ladder bitmap (pfac_context.h: struct Filter) until the candidate meets an
         * S node (walk), or neither an S nor a G node (its result is 0).  Survivors -> walk queue.  A batch takes up
         * to 64 candidates, fewer if the queue has less room (then at least 16, or all that are left). */
        for (;;) {
            const uint32_t left = listEnd - listAt;
            const uint32_t want = left < 64u ? left : 64u;
            const uint32_t room = kQCap - (qv - qh) - pendCount;    /* (pendCount: the candidates of an earlier batch of this trip that tailResolve may still append) */
            const uint32_t take = room < want ? room : want;
            if (left == 0 || (take != want && take < kAppendMin)) break;
            if (!REDUCE && left < kMergeMin && __ballot(hits != 0) != 0) break;      /* more of this chunk's hits are about to join them (step 4 of the next trip) */
            const bool act = (uint32_t)lane < take;
            const uint32_t code = act ? listCode(listAt) : 0u;
            const uint32_t o = ((code & 0x10u) << 6) | ((code >> 1) & 0x3F0u) | (code & 0xFu);
            const uint32_t at = o >> 2, sh = o & 3u;
            /* the candidate's 20 bytes and the rolling hashes of all its prefixes first: nothing in them depends on the outcome
             * of a level, so the probes of several levels are in flight together (the kernel has the registers for it since
             * the full-result variant runs one walk per lane; the compacted-output variant tests no level here) */
            const uint32_t e0 = stage[at], e1 = stage[at + 1], e2 = stage[at + 2], e3 = stage[at + 3], e4 = stage[at + 4], e5 = stage[at + 5];
            const uint32_t x0 = __builtin_amdgcn_alignbyte(e1, e0, sh), x1 = __builtin_amdgcn_alignbyte(e2, e1, sh), x2 = __builtin_amdgcn_alignbyte(e3, e2, sh),
                           x3 = __builtin_amdgcn_alignbyte(e4, e3, sh), x4 = __builtin_amdgcn_alignbyte(e5, e4, sh);
            uint32_t walk = act ? (code >> 15) & 1u : 0u;
            uint32_t und = act ? walk ^ 1u : 0u;                    /* undecided: a G node so far */
            /* Input that follows the patterns deeper than the ladder looks (near misses of long patterns: BASELINE config 5)
             * passes every level: the ladder then only costs.  A wave whose last four batches each spared less than an eighth
             * of their undecided candidates walks the next 28 batches' candidates untested, then looks again. */
            const uint32_t und0 = (uint32_t)__popcll(__ballot(und != 0));
            const bool skipLadder = !REDUCE && ladderSkip != 0;
            const bool deepLadder = a.ladderLast > pfac::kLadderLast;       /* wave-uniform (a kernel argument) */
            uint32_t stopHash = 0;                                          /* kVeto: the ladder hash of the level that told this candidate to stop */
            (void)stopHash;
            if (skipLadder) { ladderSkip--; walk |= und; und = 0; }
            if (!REDUCE && __ballot(und != 0) != 0) {
                uint32_t hl[pfac::kLadderLevels];
                hl[0] = (x0 * pfac::kLadMul0) ^ a.ladderSalt;           /* pfac::ladderStart */
#pragma unroll
                for (int lv = 1; lv < pfac::kLadderLevels; lv++) {
                    const uint32_t xw = lv <= 2 ? x1 : lv <= 4 ? x2 : lv <= 6 ? x3 : x4;
                    hl[lv] = (hl[lv - 1] ^ ((lv & 1) ? (xw & 0xFFFFu) : (xw >> 16))) * pfac::kLadMul;
                }
                /* Skip tags (pfac::Filter): a candidate whose depth-6 hash is the tag of a single path down to kLadderLast is next asked at that level
                 * (its hash there covers every byte on the way).  When the batch has tags -- a kernel argument: pattern sets with a long shared prefix,
                 * BASELINE config 5 -- and nobody else is undecided behind depth 6, the wave goes from level 6 straight to the last one: six levels of
                 * three probes each were two thirds of a batch's instructions on the near-miss stream. */
                uint32_t skipping = 0;                                          /* 0 / 1 */
                bool straightToLast = false;                                    /* wave-uniform */
                if (a.skipCount != 0) {
                    for (int k = 0; k < a.skipCount; k++) skipping |= hl[1] == a.skipTags[k] ? 1u : 0u;
                    skipping &= und;
                }
#pragma unroll
                for (int lv = 1; lv < pfac::kLadderLevels; lv++) {
                    /* one early exit, in the middle: a check per level makes every level wait for the LDS reads of the one before
                     * it, and on text a batch almost always has a candidate that follows some long keyword to the last levels */
                    if (lv == 5 && __ballot(und != 0) == 0) break;
                    /* (looked at behind levels 6, 8 and 10: the few candidates of OTHER patterns in a batch -- a Snort-scale set over the same stream -- are
                     * decided by then, as a rule) */
                    if (lv >= 2 && lv <= 4 && a.skipCount != 0 && !straightToLast) straightToLast = __ballot((und & ~skipping) != 0) == 0;
                    if (lv >= 2 && lv < pfac::kLadderLevels - 1 && straightToLast) continue;
                    /* who is asked at this level: the undecided, but for those on a tagged path between its first and its last level */
                    const uint32_t act = (lv >= 2 && lv < pfac::kLadderLevels - 1) ? (und & ~skipping) : und;
                    const uint32_t h = hl[lv];
                    const uint32_t w = ladWord(h);
                    const uint32_t sHit = ladStop(w, h);                            /* bit 0; und is 0 or 1 */
                    if constexpr (kVeto) stopHash = (act & sHit) ? h : stopHash;
                    walk |= act & sHit;
                    if (lv == pfac::kLadderLevels - 1 && !deepLadder) und = 0;      /* the last level has S nodes only */
                    else und &= ~act | (ladGoOn(w, h) & ~sHit);
                }
                if constexpr (kVeto) {
                    /* A deep ladder (pfac::Filter::ladderLast) goes on behind the 20 bytes wherever patterns still share a path: two more bytes of
                     * the staged chunk per level, rolled (BASELINE config 5: the 24-byte prefix its patterns share). */
                    if (deepLadder && __ballot(und != 0) != 0) {
                        uint32_t h = hl[pfac::kLadderLevels - 1];
                        const uint32_t staged = (uint32_t)kChunkBytes + 4u * (uint32_t)kHaloDwords;
                        /* eight bytes (four levels) of the stage at a time, read before they are needed: a level is then one LDS round trip (its probes) */
                        for (uint32_t dd = (uint32_t)pfac::kLadderLast; dd < (uint32_t)a.ladderLast && __ballot(und != 0) != 0; dd += 8u) {
                            const bool any = und != 0u && o + dd + 12u <= staged;          /* staged? (else: undecided -> walk) */
                            walk |= any ? 0u : und;
                            und = any ? und : 0u;
                            const uint32_t b = any ? o + dd : 0u;
                            const uint32_t w0 = stage[b >> 2], w1 = stage[(b >> 2) + 1u], w2 = stage[(b >> 2) + 2u];
                            const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, b & 3u), hi = __builtin_amdgcn_alignbyte(w2, w1, b & 3u);
                            uint32_t hk[4], wk[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const uint32_t piece = ((k < 2 ? lo : hi) >> (16 * (k & 1))) & 0xFFFFu;
                                h = (h ^ piece) * pfac::kLadMul;
                                hk[k] = h;
                            }
#pragma unroll
                            for (int k = 0; k < 4; k++) wk[k] = ladWord(hk[k]);
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const bool lvl = dd + 2u * (uint32_t)k < (uint32_t)a.ladderLast;        /* wave-uniform */
                                const uint32_t sHit = ladStop(wk[k], hk[k]);
                                stopHash = (lvl && (und & sHit & 1u)) ? hk[k] : stopHash;
                                walk |= lvl ? (und & sHit) : 0u;
                                und = lvl ? (und & ladGoOn(wk[k], hk[k]) & ~sHit) : und;
                            }
                        }
                    }
                }
                walk |= und;                                    /* deep ladder: undecided behind the last level looked at (kernels without VETO: kLadderLast) */
                if constexpr (kVeto) {
                    /* The tail table (pfac::Filter): a stop node below which one pattern is left knows the hash of the rest of that pattern.  A
                     * candidate that was told to stop there rolls its own hash over as many of its bytes and walks only if the two agree. */
                    const bool ask = a.tail != nullptr && stopHash != 0u && (walk & 1u) != 0u;
                    if constexpr (kVetoG) {
                        /* device memory: the bucket's two entries with ONE gathered load -- whose answer is looked at at the top of the NEXT trip, behind
                         * the loop's one wait (tailResolve): waited for here, the wait would also be for the chunk just prefetched (the counter is
                         * in-order: 1.47 -> 1.93 ms on the near-miss stream).  One set of answers can be under way: a second batch of the same
                         * trip, and a batch with fewer than kTailAskMin stopped candidates (text), walk theirs as before. */
                        if (!pendAny && (uint32_t)__popcll(__ballot(ask)) >= kTailAskMin) {
                            pendCode = ask ? (0x80000000u | o) : 0u;          /* (the loads themselves: behind the batches of this trip, once) */
                            pendHash = stopHash;
                            pendAny = true;
                            pendCount = (uint32_t)__popcll(__ballot(ask));
                            walk = ask ? 0u : walk;                 /* decided, and appended if it stands, by tailResolve */
                        }
                    } else if (__ballot(ask) != 0) {
                        const uint32_t *sTail = sHotAll;
                        const uint32_t tshift = 32u - (uint32_t)a.log2Tail;
                        const uint32_t s1 = ((uint32_t)(stopHash * pfac::kTailMul) >> tshift) * 3u, s2 = ((uint32_t)(stopHash * pfac::kTailMul2) >> tshift) * 3u;
                        const uint32_t t1 = sTail[s1], h1 = sTail[s1 + 1], i1 = sTail[s1 + 2], t2 = sTail[s2], h2 = sTail[s2 + 1], i2 = sTail[s2 + 2];
                        const bool m1 = i1 != 0u && t1 == stopHash, m2 = i2 != 0u && t2 == stopHash;
                        const uint32_t want = m1 ? h1 : h2, info = m1 ? i1 : (m2 ? i2 : 0u);
                        const uint32_t nb = info & 0xFFu, from = o + (info >> 8);
                        const bool chk = ask && info != 0u && from + nb + 4u <= (uint32_t)kChunkBytes + 4u * (uint32_t)kHaloDwords;   /* the bytes must be staged */
                        const bool vetoed = chk && tailRun(stopHash, chk, from, nb) != want;
                        if (vetoed) walk = 0u;
                        chunkEvents += (uint32_t)__popcll(__ballot(vetoed));      /* near misses: the wave's evidence of what its stream is (the tiled kernel's table follows the verdict) */
                    }
                }
            }
            if (REDUCE) walk |= und;                                /* undecided after the last level tested: walk */
            if (!REDUCE && !skipLadder && und0 >= 16u) {
                const uint32_t walked = (uint32_t)__popcll(__ballot(walk != 0)) - (take - und0);           /* of the und0 that were undecided (the other take - und0 walk anyway) */
                ladderIdle = (und0 - walked) * 8u < und0 ? ladderIdle + 1u : 0u;
                if (ladderIdle >= 4u) { ladderIdle = 0; ladderSkip = 28; }
            }
#if PFAC_ABLATE >= 3          /* timing experiment: walk only a fraction of the candidates (results are wrong) */
            walk = (((o * 2654435761u) >> 28) < (PFAC_ABLATE - 2) * 4u) ? walk : 0u;
#endif
            const bool keep = walk != 0;
            const uint64_t keepMask = __ballot(keep);
            if (keep) {
                const uint32_t qi = (qv + laneRankIn(keepMask)) & kMask;
                if constexpr (kStageWalk) {
                    queue32[qi] = (cur << 31) | o;             /* the walk reads its input from the stage (text mode: copies its first bytes when it starts) */
                } else {
                const u32x4 entry = {stagedBase + o, x0, x1, x2};
                const u32x2 entryB = {x3, x4};
                queue[qi] = entry;
                queueB[qi] = entryB;
                if (kWideEntry) {                               /* bytes 20..35: read now, for the few that are kept */
                    const uint32_t e6 = stage[at + 6], e7 = stage[at + 7], e8 = stage[at + 8], e9 = stage[at + 9];
                    const u32x4 entryC = {__builtin_amdgcn_alignbyte(e6, e5, sh), __builtin_amdgcn_alignbyte(e7, e6, sh),
                                          __builtin_amdgcn_alignbyte(e8, e7, sh), __builtin_amdgcn_alignbyte(e9, e8, sh)};
                    queueC[qi] = entryC;
                }
                }
            }
            qv = uni(qv + (uint32_t)__popcll(keepMask));
            listAt = uni(listAt + take);
        }
        if constexpr (kVetoG) {
            /* the buckets of the batch that asked: ONE load site per trip, behind the loop over the batches (inside it the compiler would wait for the
             * load of a batch before -- there is none: pendAny -- before it reuses the registers) */
            if (pendAny && (pendCode >> 31) != 0u)
                pendE = reinterpret_cast<const u32x4 *>(a.tail)[(uint32_t)(pendHash * pfac::kTailMul) >> (32u - (uint32_t)a.log2Tail)];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        PFAC_TICK(5);
#if PFAC_ABLATE == 2
        qh = qv;                                        /* timing experiment: drop the verified entries unwalked */
#endif
    }
#if PFAC_ABLATE == 1
    if (ablateSink == 0x12345u) a.out[0] = 1;
#endif
    if (REDUCE || kStagedPatch) flushStaged();
    if (!REDUCE && a.denseList != nullptr) flushDense();
#if PFAC_TIMING
    if (lane == 0)
        for (int k = 0; k < 12; k++) atomicAdd(reinterpret_cast<unsigned long long *>(a.work + pfac::kStatsWord) + 8 + k, (unsigned long long)tm[k]);
#endif
    }   /* scanning wave */

    /* counters of this launch (PFACX_getScanStats): per-wave scalars -> LDS -> one atomic per counter and block */
    __syncthreads();
    if (tid < 8) sGram3[tid] = 0;
    __syncthreads();
    if (lane == 0) {
        atomicAdd(&sGram3[0], stRounds); atomicAdd(&sGram3[1], stLaneSteps);
        atomicAdd(&sGram3[2], stStarts); atomicAdd(&sGram3[3], stHits); atomicAdd(&sGram3[4], stCand);
        if (!REDUCE && advBalance > 0 && !(kWriters && wave >= kScanners)) atomicAdd(&sGram3[5], 1u);
    }
    __syncthreads();
    if (tid < 4) atomicAdd(reinterpret_cast<unsigned long long *>(a.work + pfac::kStatsWord) + tid, (unsigned long long)sGram3[tid]);
    if (tid == 5) atomicAdd(reinterpret_cast<unsigned long long *>(a.work + pfac::kStatsWord) + 5, (unsigned long long)sGram3[4]);
    if (!REDUCE && tid == 6 && sGram3[5] != 0) atomicAdd(a.work + pfac::kModeVotesWord, sGram3[5]);
    /* The last block out leaves the counters as the next launch needs them -- zero -- and publishes the statistics: a
     * memset in front of every launch was 5 us of a call (profiles/r03_experiments.md section 7).  Every block counts
     * itself out after its own atomics have been performed; whoever counts last knows that all the others are done. */
    if (wave == 0) {
        __threadfence();
        unsigned int before = 0;
        if (lane == 0) before = atomicAdd(a.work + pfac::kDoneWord, 1u);
        before = (unsigned int)__builtin_amdgcn_readfirstlane((int)before);
        if (before == gridDim.x - 1u) {
            __threadfence();
            unsigned long long *acc = reinterpret_cast<unsigned long long *>(a.work + pfac::kStatsWord);
            unsigned long long *published = reinterpret_cast<unsigned long long *>(a.work + pfac::kStatsPublishedWord);
            if (lane < 32) atomicExch(a.work + lane * 32, 0u);                              /* the parts' claim counters */
            if (lane < pfac::kStatsCount) published[lane] = lane == 4 ? (unsigned long long)a.n : atomicExch(acc + lane, 0ull);
            if (lane == pfac::kStatsCount) {
                const unsigned int denseChunks = REDUCE ? 0u : atomicAdd(a.work + a.denseWord, 0u);   /* stays: the tiled kernel behind this launch reads it */
                published[lane] = (unsigned long long)denseChunks;
                /* most chunks pattern-dense: the handle's next big call goes to the tiled kernel alone (scan(): PFACX_KERNEL_AUTO), which
                 * walks such input in place and reports in turn when the stream stops being dense */
                if (!REDUCE && a.hostHint != nullptr) __hip_atomic_store(a.hostHint + 1, denseChunks * 2u > numChunks ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (lane == pfac::kStatsCount + 1) published[lane] = (unsigned long long)kWalkSets;
            if (lane == pfac::kStatsCount + 2) {                  /* scanning waves that ended the launch in stage mode: published, and the next launch's starting mode */
                const unsigned int votes = !REDUCE ? atomicExch(a.work + pfac::kModeVotesWord, 0u) : 0u;
                published[lane] = (unsigned long long)votes | ((unsigned long long)(kStageWalk ? 1u : 0u) << 32) | ((unsigned long long)(kVeto ? (kVetoG ? 2u : 1u) : 0u) << 33);
                if (!REDUCE) {
                    const unsigned int hint = votes * 2u >= gridDim.x * (unsigned int)kScanners ? 1u : 0u;
                    atomicExch(a.work + pfac::kModeHintWord, hint);
                    /* ... and where the host sees it without asking (host memory): which walker the handle's next launch gets */
                    if (a.hostHint != nullptr) __hip_atomic_store(a.hostHint, hint, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            if (lane == 32) atomicExch(a.work + a.denseWordOther, 0u);
            if (lane == 33) atomicExch(a.work + pfac::kDoneWord, 0u);
        }
    }
#if PFAC_STATS
    if (lane == 0 && (blockIdx.x % 32) == 0 && wave == 0)
        printf("STATS block %d wave0 fullRounds %u slotGathers %u winLoads %u startDead %u\n", (int)blockIdx.x, stFullRounds, stSlotGathers, stWinLoads, stStartDead);
#endif
}

/* ------------------------------------------------------------- launching */

/* the CU's 160 KiB: the prefilter bitmaps (<= kFilterLdsBudget, pattern_compiler.cpp) + control block + per scanning wave a
 * walk queue (24 B per entry), the staged chunk and the hit list (+ the pair staging of the compacted-output variant) */
constexpr size_t kScannerLdsStage = kQueueCap * 4 + 2 * kWalkStageBytes, kScannerLdsWindow = kQueueCap * (4 + kEntryBytesFull) + kStageWords * 4;
constexpr size_t kScannerLdsFull = (size_t)(kWavesPerBlock - PFAC_WRITERS) * ((kScannerLdsStage > kScannerLdsWindow ? kScannerLdsStage : kScannerLdsWindow) +
                                                                              kListCap * 2 + (kStagedPatch ? kReduceCap * 8 : 0) + kDenseStage * 4);
constexpr size_t kScannerLdsReduce = (size_t)kReduceScanners * (kReduceQueueCap * 24 + kStageWords * 4 + kListCap * 2 + kReduceCap * 8);
static_assert(pfac::kFilterLdsBudget + kControlWords * 4 + kScannerLdsFull <= kLdsPerCu, "prefilter bitmaps + scanning waves' buffers must fit the CU's LDS");
static_assert(kGram1LdsOffset + kGram1LdsBytes + 1024 /* final3 */ + 8192 /* 2-byte bitmap */ + kControlWords * 4 + kScannerLdsReduce <= kLdsPerCu,
              "compacted-output kernel: gram1 + prefix4 + final3 + short bitmap + scanning waves' buffers must fit the CU's LDS");

size_t filterLdsBytes(const PFAC_context *c, bool reduce, bool stage)
{
    size_t bytes = reduce ? (size_t)kGram1LdsOffset + kGram1LdsBytes + (size_t(1) << c->filter.log2BitsF3) / 8
                          : kLadderLdsOffset + ((size_t(1) << c->filter.log2BitsLad) + (size_t(1) << c->filter.log2BitsF3)) / 8;   /* the level-1 bitmap has its 32 KiB whatever its size */
    if (c->filter.hasShort) bytes += 65536 / 8;
    const size_t scanners = reduce ? (size_t)kReduceScanners : (size_t)kWavesPerBlock - PFAC_WRITERS;
    bytes += kControlWords * sizeof(uint32_t);
    if (!reduce && stage) bytes += scanners * (kQueueCap * 4 + 2 * kWalkStageBytes + (kListCap / 2) * sizeof(uint32_t));
    else bytes += scanners * ((reduce ? kReduceQueueCap * (4 + kEntryBytes) : kQueueCap * (4 + kEntryBytesFull)) + (kStageWords + kListCap / 2) * sizeof(uint32_t));
    if (reduce || kStagedPatch) bytes += scanners * kReduceCap * 2 * sizeof(uint32_t);
    if (!reduce) bytes += scanners * kDenseStage * sizeof(uint32_t);
    return bytes;
}

/* LDS of the tail table behind the buffers of a window-walker launch, or 0: none, or no room */
size_t vetoLdsBytes(const PFAC_context *c, const ScanArgs &a)
{
    if (a.tail == nullptr || a.log2Tail < 2 || c->filter.tail.empty()) return 0;      /* (a table in its device-memory form takes no LDS: VETO = 2) */
    const size_t bytes = size_t(12) << a.log2Tail;
    return filterLdsBytes(c, false, false) + bytes <= kLdsPerCu ? bytes : 0;
}

template <bool TEX, bool HAS_SHORT, bool REDUCE, bool STAGE, int VETO = 0>
hipError_t launchFilter(const PFAC_context *c, const ScanArgs &a0)
{
    auto kernel = pfac_scan_filter<TEX, HAS_SHORT, REDUCE, REDUCE ? PFAC_WALK_SETS : PFAC_WALK_SETS_FULL, STAGE, VETO>;
    static ShapeCache cache;
    size_t lds = filterLdsBytes(c, REDUCE, STAGE) + (VETO == 1 ? vetoLdsBytes(c, a0) : 0);
    int dev = -1;                                      /* the device the launch goes to: the CURRENT one (the library never switches devices) */
    hipError_t de = hipGetDevice(&dev);
    if (de != hipSuccess) return de;
    if (lds > kLdsPerCu || dev < 0 || dev >= kMaxDevices) return hipErrorInvalidValue;
    ScanArgs a = a0;
    a.hotSlots = 0;
    if (!REDUCE && STAGE) {
        /* the LDS the bitmaps and the waves' buffers leave holds the top of the chained table (buckets breadth first, then the
         * initial state's row): 50 KiB and more for a set of a few thousand patterns, nothing for a Snort-scale set */
#ifndef PFAC_FILTER_HOT
#define PFAC_FILTER_HOT 1
#endif
        size_t hot = PFAC_FILTER_HOT ? (kLdsPerCu - lds) / sizeof(pfac::ChainSlot) : 0;
        const size_t top = (size_t)a.rootRow + (size_t)pfac::kCharSet;
        if (hot > top) hot = top;
        if (hot < 1024) hot = 0;                       /* not worth a test per step */
        a.hotSlots = (uint32_t)hot;
        lds += hot * sizeof(pfac::ChainSlot);
    }
    int perCU;
    {
        std::lock_guard<std::mutex> g(cache.lock);
        if (cache.perCU[dev] == 0) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
            if (e != hipSuccess) return e;
            int n = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, kBlockThreads, kLdsPerCu);
            if (e != hipSuccess) return e;
            cache.perCU[dev] = n < 1 ? 1 : n;
        }
        perCU = cache.perCU[dev];
    }
    const size_t numChunks = a.n / kChunkBytesHost;
    constexpr size_t scanners = REDUCE ? (size_t)kReduceScanners : (size_t)kWavesPerBlock - PFAC_WRITERS;
    size_t blocks = (numChunks + scanners - 1) / scanners;
    const size_t resident = (size_t)(c->multiProcessorCount > 0 ? c->multiProcessorCount : 256) * perCU;
    if (blocks > resident) blocks = resident;
    hipError_t e = hipSuccess;
    /* the launch counters are left zero by the launch before (see the kernel's end) -- unless that one failed; the stage
     * timers of the profile build are only ever added to */
    if (PFAC_TIMING || c->countersDirty) {
        e = hipMemsetAsync(c->d_workCounters, 0, pfac::kWorkCounterWords * sizeof(unsigned int), 0);
        if (e != hipSuccess) return e;
        c->countersDirty = false;
    }
    const bool timed = c->kernelTiming && c->evTime[0] && c->evTime[1];
    if (timed) (void)hipEventRecord(static_cast<hipEvent_t>(c->evTime[0]), 0);
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlockThreads), lds, 0, a);
    e = hipGetLastError();
    if (e != hipSuccess) c->countersDirty = true;
    if (timed) c->evTimeRecorded = hipEventRecord(static_cast<hipEvent_t>(c->evTime[1]), 0) == hipSuccess;
#if PFAC_TIMING
    {
        unsigned long long t[18];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(t, c->d_workCounters + pfac::kStatsWord + 16, sizeof(t), hipMemcpyDeviceToHost);
        double sum = 0;
        for (int k = 0; k < 12; k++) sum += (double)t[k];
        static const char *names[12] = {"consume", "refill", "issue", "level1+stage", "list-fences", "ladder+append", "resolve(wait for writers)+prefetch issue", "loop/pop/other",
                                        "wait for loads (top of trip)", "popcount+scan", "emit-loop", "level-4 test"};
        fprintf(stderr, "PFAC_TIMING blocks %zu scanners %zu:", blocks, scanners);
        for (int k = 0; k < 12; k++) fprintf(stderr, "  %s %.1f%% (%.0f cyc/wave)", names[k], 100.0 * t[k] / sum, (double)t[k] / (blocks * scanners));
        if (!REDUCE && PFAC_WRITERS) {
            static const char *wnames[6] = {"wait for the other writer's claim", "wait for run-ahead room (scanners)", "device claim (atomic)", "issue zero stores",
                                            "wait until the zeros are in L2", "publish in order"};
            double wsum = 0;
            for (int k = 0; k < 6; k++) wsum += (double)t[12 + k];
            fprintf(stderr, "\nPFAC_TIMING writers %d per block:", (int)PFAC_WRITERS);
            for (int k = 0; k < 6; k++) fprintf(stderr, "  %s %.1f%% (%.0f cyc/wave)", wnames[k], 100.0 * t[12 + k] / wsum, (double)t[12 + k] / (blocks * PFAC_WRITERS));
        }
        fprintf(stderr, "\n");
    }
#endif
    return e;
}

/* the filter kernel walks the chained table in both perf modes; "texture" = buffer-resource loads */
template <bool REDUCE>
hipError_t launchChained(const PFAC_context *c, const ScanArgs &a, bool tex)
{
    /* the full-result kernel's walker (PFACX_setWalker): by default what the handle's last full-result launch found -- near
     * misses all over (most of its scanning waves ended it in stage mode / expecting long slots) -> StageLane, text -> the
     * register-window walker.  The word is host memory the last block of a launch writes: nothing is waited for, a launch
     * still under way simply has not voted yet */
    bool stage = false;
    int veto = 0;
    if (!REDUCE) {
        /* a pattern set with a tail table (a few thousand patterns: its tables leave the LDS for it) puts the ladder's stops to the table
         * before they become walks: near misses hardly reach a walker then, and the register-window walker is the one for few walks */
#ifndef PFAC_VETO
#define PFAC_VETO 1
#endif
        const bool nearMisses = c->h_modeHint != nullptr && *static_cast<volatile const unsigned int *>(c->h_modeHint) != 0;      /* the handle's last full-result launch */
        if (PFAC_VETO && c->walker != PFACX_WALKER_STAGE) {
            if (vetoLdsBytes(c, a) != 0) veto = 1;
            /* ... in device memory (Snort-scale sets): a memory round trip per batch of stopped candidates, which only a stream of near misses repays:
             * the veto kernel takes the place of the stage walker, and goes on reporting near misses (the candidates it vetoes) while the stream stays so */
            else if (a.tail != nullptr && !c->filter.tailG.empty() && (c->walker == PFACX_WALKER_VETO || (c->walker == PFACX_WALKER_AUTO && nearMisses))) veto = 2;
        }
        stage = c->walker == PFACX_WALKER_STAGE || (veto == 0 && c->walker == PFACX_WALKER_AUTO && nearMisses);
    }
#ifdef PFAC_QUICK      /* development builds (register / ISA inspection): the bench instances only */
    if (REDUCE || !tex) return hipErrorNotSupported;
    if (stage) return c->filter.hasShort ? launchFilter<true, true, false, true>(c, a) : launchFilter<true, false, false, true>(c, a);
    if (veto == 2) return c->filter.hasShort ? launchFilter<true, true, false, false, 2>(c, a) : launchFilter<true, false, false, false, 2>(c, a);
    if (veto) return c->filter.hasShort ? launchFilter<true, true, false, false, 1>(c, a) : launchFilter<true, false, false, false, 1>(c, a);
    return c->filter.hasShort ? launchFilter<true, true, false, false>(c, a) : launchFilter<true, false, false, false>(c, a);
#else
    if (!REDUCE && veto == 2) {
        if (tex) return c->filter.hasShort ? launchFilter<true, true, false, false, 2>(c, a) : launchFilter<true, false, false, false, 2>(c, a);
        return c->filter.hasShort ? launchFilter<false, true, false, false, 2>(c, a) : launchFilter<false, false, false, false, 2>(c, a);
    }
    if (!REDUCE && veto) {
        if (tex) return c->filter.hasShort ? launchFilter<true, true, false, false, 1>(c, a) : launchFilter<true, false, false, false, 1>(c, a);
        return c->filter.hasShort ? launchFilter<false, true, false, false, 1>(c, a) : launchFilter<false, false, false, false, 1>(c, a);
    }
    if (!REDUCE && stage) {
        if (tex) return c->filter.hasShort ? launchFilter<true, true, false, true>(c, a) : launchFilter<true, false, false, true>(c, a);
        return c->filter.hasShort ? launchFilter<false, true, false, true>(c, a) : launchFilter<false, false, false, true>(c, a);
    }
    if (tex) return c->filter.hasShort ? launchFilter<true, true, REDUCE, false>(c, a) : launchFilter<true, false, REDUCE, false>(c, a);
    return c->filter.hasShort ? launchFilter<false, true, REDUCE, false>(c, a) : launchFilter<false, false, REDUCE, false>(c, a);
#endif
}

} // namespace

namespace pfacmod {
hipError_t launchFilterKernel(const PFAC_context *c, const ScanArgs &a, bool tex, bool reduce)
{
    return reduce ? launchChained<true>(c, a, tex) : launchChained<false>(c, a, tex);
}
}

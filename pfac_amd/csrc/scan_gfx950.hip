/*
 * scan_gfx950.hip -- kernel module libpfac_gfx950.so: the PFAC match path for
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
 *      counts, one divergent loop).  A chunk in which more than 90 % of the positions
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
#error "scan_gfx950.hip is written for gfx950 (CDNA4): wave64, gfx9 waitcnt semantics, 160 KiB LDS"
#endif
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include <cstdint>
#include <type_traits>
#include <vector>

#include "pfac_context.h"

namespace {

using pfac::Int2;

constexpr int kTrap = pfac::kTrapState;
#ifndef PFAC_BLOCK_THREADS
#define PFAC_BLOCK_THREADS 1024
#endif
#ifndef PFAC_MIN_WAVES_PER_SIMD
#define PFAC_MIN_WAVES_PER_SIMD 1                    /* HIP: second __launch_bounds__ argument = minimum waves per SIMD */
#endif
#ifndef PFAC_SLOT_AUX
#define PFAC_SLOT_AUX 0                         /* cache policy of the walkers' slot loads (buffer path): 1 = sc0, 2 = nt, 16 = sc1 */
#endif
#ifndef PFAC_QUEUE_CAP
#define PFAC_QUEUE_CAP 64
#endif
constexpr int kBlockThreads = PFAC_BLOCK_THREADS;
constexpr int kWavesPerBlock = kBlockThreads / 64;
constexpr int kTileBytes = 1024;              /* input bytes one wave-wide 16 B/lane load covers */
constexpr uint32_t kLadderLdsOffset = (uint32_t)pfac::kGram3LdsBytes;  /* LDS: [0, 32 KiB) the level-1 bitmap (at most 2^18 bits), then the prefix ladder */
/* ... of the compacted-output kernel: [0, 16 KiB) the 4-byte prefixes, [16, 80 KiB) its one-bit level-1 bitmap (pfac_context.h: gram1, prefix4) */
constexpr uint32_t kPrefix4LdsBytes = (1u << pfac::kPrefix4Log2) / 8, kGram1LdsOffset = kPrefix4LdsBytes, kGram1LdsBytes = (1u << pfac::kGram1Log2) / 8;

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum TableMode { DENSE_GLOBAL = 0, DENSE_BUFFER = 1, HASH_GLOBAL = 2, HASH_BUFFER = 3 };

struct ScanArgs {
    const unsigned char *in;
    int *out;
    size_t n;                                          /* filter kernel: owned = readable bytes handled here (whole chunks); naive: readable bytes */
    size_t owned;                                      /* tiled / naive kernel: positions [0, owned) get a result */
    const int *dense;
    const Int2 *hashRow;
    const Int2 *hashVal;
    const u32x4 *chainSlots;                           /* pfac::ChainSlot[], 16 bytes each               */
    uint32_t extDelta;                                 /* the extension unit of slot i is chainSlots[i + extDelta] (long slots of wide buckets: pfac_context.h) */
    uint32_t jumpLongBase;                             /* the long jump table (same hash, slots with chains of up to kChainMaxWide bytes) */
    uint32_t rootRow, jumpBase, jumpShift;             /* inside chainSlots: the initial state's bucket (256 slots, indexed by the byte)
                                                          and the jump table (2^(32 - jumpShift) slots): tables.cpp */
    uint32_t denseBytes, hashRowBytes, hashValBytes, chainBytes;     /* buffer-resource extents */
    uint32_t maxWalk;                                  /* longest pattern: no walk reads further from its start position */
    uint32_t hotSlots;                                 /* tiled kernel, and the full-result filter kernel with what LDS its bitmaps and buffers leave (small pattern sets):
                                                          the first hotSlots slots of chainSlots (the buckets the initial state's transitions land in,
                                                          breadth first) are copied to LDS by every block */
    const int *initialRow;
    const uint32_t *gram3;
    const uint32_t *gram1, *prefix4;                   /* compacted-output kernel: its level 1 and depth-4 test */
    const uint32_t *ladder;
    const uint32_t *final3;
    const uint32_t *shortBits;
    int log2Bits, log2BitsLad, log2BitsF3;
    int numFinal;
    int initialState;
    unsigned int *work;                                /* pfac::kWorkCounterWords zeroed counters: next chunk of each input part */
    unsigned int *hostHint;                            /* host memory (mapped): [0] 1 = most scanning waves of this full-result launch found their stream full of near misses;
                                                          [1] 1 = most of the launch's chunks (tiled kernel: groups) were pattern-dense */
    uint32_t reportDense;                              /* tiled kernel: this launch is a whole call: report [1] */
    /* compacted output (PFAC_matchFromDeviceReduce): unordered append, sorted by position afterwards */
    int *reducePos;
    unsigned int *reduceCount;
    unsigned int reduceBase;                           /* position of a.in[0] inside the caller's stream */
    /* pattern-dense chunks (full-result path): the filter kernel lists the chunks in which most positions pass level 1
     * instead of filtering them; the tiled kernel that follows it
     * walks their positions one per thread.  denseIn / denseOut / denseReadable describe the filter launch the chunk
     * numbers refer to.  The list's length is a.work[kDenseCountWord]; a wave appends 8 chunks at a time (one device
     * counter answers ~90 atomics per microsecond: an append per chunk cost 1.5 ms for 256 MiB of pattern-dense input). */
    unsigned int *denseList;
    uint32_t denseWord, denseWordOther;                /* a.work[denseWord] counts this launch's dense chunks; the other one is left zero for the next launch */
    const unsigned char *denseIn;
    int *denseOut;
    size_t denseReadable;
    /* the ends of the input, which the filter kernel's unchecked loads must not come near (the <= 15 positions in front
     * of the first 16-byte aligned byte, and the last partial chunk + maxPatternLen + 64 bytes): positions [endsA0, endsA1)
     * and [endsB0, endsB1) of endsIn (endsReadable bytes can be read), results to endsOut -- walked with bounds, one
     * position per lane, by the first scanning wave of the first blocks BEFORE it starts scanning, so that their chain
     * of dependent loads (15 us as a launch of its own behind the filter kernel: 2 % of a call) hides behind the scan */
    const unsigned char *endsIn;
    int *endsOut;
    size_t endsReadable;
    uint32_t endsA0, endsA1, endsB0, endsB1;
};
using pfac::kDenseCountWord;                            /* the launch counters are one 128-byte line each: lines 0..31 hand out the input (at most 32 parts),
                                                          line 32 or 34 (ScanArgs::denseWord) counts the dense chunks */
constexpr uint32_t kDenseStage = 8;                     /* dense chunks a wave collects in LDS before it appends them to the list */
#ifndef PFAC_DENSE_HITS
#define PFAC_DENSE_HITS 1024
#endif
constexpr uint32_t kDenseHits = PFAC_DENSE_HITS;       /* of the 2048 positions of a chunk: above half, listing, testing and queueing the survivors costs more than the
                                                          tiled kernel's dense mode, which walks every position of such a chunk in place (round 3, with the
                                                          reference-shaped kernel behind the list, needed 90 %: profiles/r03_experiments.md) */

/* ---------------------------------------------------------------- lookups */

/* One automaton transition beyond the initial state.
 * ref dense:  *(d_PFAC_table + state*CHAR_SET + inputChar), PFAC_kernel.cu:291
 * ref hashed: notex_lookup / tex_lookup, PFAC_kernel_spaceDriven.cu:76-124   */
template <int MODE> struct Lookup;

template <> struct Lookup<DENSE_GLOBAL> {
    const int *table;
    __device__ explicit Lookup(const ScanArgs &a) : table(a.dense) {}
    __device__ __forceinline__ int operator()(int state, int ch) const
    {
        return table[(size_t)(uint32_t)state * pfac::kCharSet + (uint32_t)ch];
    }
};

/* "texture" analogue: read-only, bounds-checked buffer resource (out-of-range
 * reads return 0, the hardware counterpart of cudaAddressModeClamp at
 * PFAC_kernel.cu:126-129; state 0 is the unused all-trap row). */
template <> struct Lookup<DENSE_BUFFER> {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ explicit Lookup(const ScanArgs &a)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(a.dense), 0, (int)a.denseBytes, 0x00020000)) {}
    __device__ __forceinline__ int operator()(int state, int ch) const
    {
        const uint32_t off = ((uint32_t)state * pfac::kCharSet + (uint32_t)ch) * 4u;
        return (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
    }
};

__device__ __forceinline__ int hashSlot(int kAndMask, int ch)
{
    /* (k*ch) mod 257 without a divide: 256 == -1 (mod 257), k*ch <= 65280 */
    const int x = (kAndMask >> 16) * ch;
    int r = (x & 0xFF) - (x >> 8);
    r += (r < 0) ? pfac::kHashP : 0;
    return r & (kAndMask & 0xFFFF);
}

template <> struct Lookup<HASH_GLOBAL> {
    const Int2 *rowPtr;
    const Int2 *valPtr;
    __device__ explicit Lookup(const ScanArgs &a) : rowPtr(a.hashRow), valPtr(a.hashVal) {}
    __device__ __forceinline__ int operator()(int state, int ch) const
    {
        const Int2 r = rowPtr[(uint32_t)state];
        if (r.x < 0) return kTrap;
        const Int2 v = valPtr[(uint32_t)r.x + (uint32_t)hashSlot(r.y, ch)];
        return v.y == ch ? v.x : kTrap;
    }
};

template <> struct Lookup<HASH_BUFFER> {
    __amdgpu_buffer_rsrc_t rowRsrc, valRsrc;
    __device__ explicit Lookup(const ScanArgs &a)
        : rowRsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<Int2 *>(a.hashRow), 0, (int)a.hashRowBytes, 0x00020000)),
          valRsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<Int2 *>(a.hashVal), 0, (int)a.hashValBytes, 0x00020000)) {}
    __device__ __forceinline__ int operator()(int state, int ch) const
    {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rowRsrc, (int)((uint32_t)state * 8u), 0, 0);
        if ((int)r.x < 0) return kTrap;
        const uint32_t slot = r.x + (uint32_t)hashSlot((int)r.y, ch);
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(valRsrc, (int)(slot * 8u), 0, 0);
        return (int)v.y == ch ? (int)v.x : kTrap;
    }
};

/* ------------------------------------------------------------------ walkers */

constexpr uint32_t kQueueCap = PFAC_QUEUE_CAP;           /* ring entries per wave (power of two)                 */

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ uint32_t laneRankIn(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t testBit(const uint32_t *bitmap, uint32_t h) { return (bitmap[h >> 5] >> (h & 31)) & 1u; }

/* LDS view of one block */
struct Lds {
    const uint32_t *gram3, *ladder, *final3, *shortBits;
    uint32_t shift3, shiftLad, shiftF3;
};

/* 16 input bytes from the 4-byte aligned address at or below byte `pos`.  No bound: the launcher only
 * gives this kernel positions whose walks end at least 32 bytes before the end of the input. */
__device__ __forceinline__ u32x4 loadWindow16(const uint32_t *in32, uint32_t pos)
{
    return *reinterpret_cast<const u32x4_a4 *>(in32 + (pos >> 2));
}

/*
 * Walkers are split-phase: issue() starts the load of the next transition, consume() finishes it.
 * Each lane runs kWalkSets independent walks; all of them issue at the top of a trip of the scan loop
 * and are consumed at the top of the next one, so one memory round trip covers up to 64 x kWalkSets
 * table steps and hides behind a whole chunk of filter work.
 *
 * What is scarce on pattern-dense input is gathered loads, then instruction issue:
 * a step is written as straight-line selects (every early `return` costs exec-mask bookkeeping for
 * the whole wave), positions are 32-bit, and nothing checks a bound -- the launcher hands the last
 * maxPatternLen + 64 bytes of the input to bounds-checked walks (ScanArgs::endsIn), so a walk that starts in this kernel's
 * range can neither run past the input nor load past it.
 *
 * Both table modes walk the CHAINED table (tables.cpp: buildChainedHashTable): a device-only copy of the
 * reference's hashed table with 16-byte slots.  A step consumes the edge byte plus the slot's
 * single-successor chain (up to 7 bytes) with one dependent memory round trip and ONE gathered load (the
 * reference's dense walk needs one per byte, its hashed walk two: PFAC_kernel.cu:255-299,
 * PFAC_kernel_spaceDriven.cu:76-124).  Gathered loads that miss the L1 cost ~2.3 cycles per lane on a
 * CU whatever their size (tools/gather_probe.hip), hence the packed slot.  The input comes with the
 * walk: the queue entry carries the 20 bytes from the start position, which is where 99.9 % of the walks
 * of the Snort-style workload end (84 % within 16, 62 % within 12); only a walk that outruns them loads
 * input, 16 bytes at a time.
 */
template <bool TEX> struct ChainCtx {
    const u32x4 *slots;
    __amdgpu_buffer_rsrc_t rsrc;
    const uint32_t *in32;
    uint32_t rootRow, jumpBase, jumpShift, extDelta, jumpLongBase;
    uint32_t hotAddr = 0, hotSlots = 0;                /* StageLane: the first hotSlots slot headers are in LDS at byte address hotAddr */
    __device__ ChainCtx(const ScanArgs &a)
        : slots(a.chainSlots),
          rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(a.chainSlots), 0, (int)a.chainBytes, 0x00020000)),
          in32(reinterpret_cast<const uint32_t *>(a.in)), rootRow(a.rootRow), jumpBase(a.jumpBase), jumpShift(a.jumpShift), extDelta(a.extDelta), jumpLongBase(a.jumpLongBase) {}
};
constexpr uint32_t kRootKs = pfac::kChainRootMeta;      /* the initial state's bucket: k = 128, S = 256 -- the slot of byte b is b */

/* slot of edge byte ch in the bucket described by the meta word `ks` of the slot that led to it (k = bits 16..23,
 * S - 1 = bits 24..31): ((k * ch) >> 7) & (S - 1) -- pfac::chainSlotOf.  The reference's family, ((k*ch) mod 257) & (S-1)
 * (PFAC_kernel_spaceDriven.cu:76-124), costs nine instructions a step without a divide; this one four, and a walk
 * step is all instruction issue once the rows are in LDS or L2. */
__device__ __forceinline__ uint32_t chainHashSlot(uint32_t ks, uint32_t ch)
{
    return ((uint32_t)__umul24(__builtin_amdgcn_ubfe(ks, 16u, 8u), ch) >> 7) & (ks >> 24);
}

constexpr uint32_t kEntryBytes = 20;           /* input bytes a queue entry brings along: compacted-output kernel (what the prefix ladder looks at) */
constexpr uint32_t kEntryBytesFull = 36;       /* ... full-result kernel: 16 more, so that a walk 21..36 bytes deep (near misses of long patterns) needs no
                                                * gathered input load: those were 40 % of the gathered loads of BASELINE config 5 */
#ifndef PFAC_WIDE_SPEC
#define PFAC_WIDE_SPEC 0                       /* register-window walker of a full-result kernel: 1 = the extension unit of a wide bucket's slot can be fetched WITH
                                                * the header (four more registers per lane, 0.5 % of the text stream's launch time); 0 = fetched when a header's first
                                                * 8 chain bytes have matched, and waited for on the spot.  The window walker is the one for text (a stream full of
                                                * near misses gets the stage walker from its second launch on: launchChained), so it does not speculate */
#endif

__device__ __forceinline__ uint32_t slotLen(uint32_t meta) { return __builtin_amdgcn_ubfe(meta, pfac::kSlotLenShift, 5u); }
/* the low n (0..8) bytes of d are zero */
__device__ __forceinline__ bool lowBytesZero(uint64_t d, uint32_t n) { return n >= 8u ? d == 0 : ((d << 8) << (56u - 8u * n)) == 0; }
/* byte i (0..15) of the 16 bytes y0..y3: v_perm_b32 takes its byte selector from a register */
__device__ __forceinline__ uint32_t byteOf16(uint32_t y0, uint32_t y1, uint32_t y2, uint32_t y3, uint32_t i)
{
    const bool up = (i & 8u) != 0;
    return __builtin_amdgcn_perm(up ? y3 : y1, up ? y2 : y0, i & 7u) & 0xFFu;
}
/* bit number of the lowest set bit, 0xFFFFFFFF if there is none (v_ffbl_b32; __builtin_ctz is undefined for 0) */
__device__ __forceinline__ uint32_t lowestBit(uint32_t x)
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
/* chain bytes 8 .. len-1 of a long slot (extension unit e) against the input bytes 8 .. 23 behind the edge byte (y0..y3), 8 <= len <= 23:
 * the first bit in which the 16 bytes differ (the dword's number folded into the bit number; no difference: 0xFFFFFFFF) must lie
 * behind the len - 8 bytes that count */
__device__ __forceinline__ bool extensionEqual(const u32x4 &e, uint32_t y0, uint32_t y1, uint32_t y2, uint32_t y3, uint32_t len)
{
    const uint32_t b0 = lowestBit(y0 ^ e.x), b1 = lowestBit(y1 ^ e.y) | 32u, b2 = lowestBit(y2 ^ e.z) | 64u, b3 = lowestBit(y3 ^ e.w) | 96u;
    const uint32_t m01 = b0 < b1 ? b0 : b1, m23 = b2 < b3 ? b2 : b3;
    return (m01 < m23 ? m01 : m23) >= 8u * (len - 8u);
}

template <bool TEX, uint32_t ENTRY> struct ChainLane {
    using Ctx = ChainCtx<TEX>;
    static constexpr bool kDeep = ENTRY > kEntryBytes;          /* 36-byte entries: the window is nine dwords, re-fetched 32 bytes at a time */
    static constexpr bool kSpec = kDeep && PFAC_WIDE_SPEC != 0; /* wide buckets: header and extension unit are fetched together, the unit compared out of the window */
    uint32_t pos = 0;
    uint32_t row = 0;                          /* first slot of the current state's bucket */
    int match = 0;
    uint32_t ks = 0, b0 = 0, depth = 0;
    /* The input window stays in registers across steps: five dwords that hold the bytes [wend - 20, wend).
     * It starts as the queue entry's 20 bytes and is only re-fetched (16 bytes into W1..W4) when a MATCHING
     * slot needs bytes beyond it: the step that ends a walk -- a slot for some other byte -- needs none. */
    u32x4 t = {0, 0, 0, 0};
    u32x4 E;                                   /* kSpec: the extension unit of the slot in flight (loaded whenever the bucket is wide, read only then: no initial value,
                                                * which would be four register writes that wait for the loads of the bounded walks in front of the scan loop) */
    uint32_t W0 = 0;
    u32x4 W = {0, 0, 0, 0};                    /* W1..W4 as one register quad: the destination of the window load itself */
    u32x4 V = {0, 0, 0, 0};                    /* kDeep: W5..W8 */
    uint32_t wend = 0;
    bool needWin = false, needSlot = true;
    bool haveE = false, needExt = false, usedE = false;   /* kSpec: the unit in E belongs to the slot in t; a long slot's header matched without it; ... and it was looked at */
    bool tookLong = false;                     /* !kSpec: this step went through a long slot (its unit fetched on the spot) */
    bool first = false;                        /* the slot in flight comes from the jump table */
    bool longWalk = false;                     /* a walk that outran its window once fetches a new one with every step from then on */

    /* input bytes q .. q+7 out of the window (q - (wend - 20) in 0..19; bytes beyond the window are garbage:
     * callers only use bytes below wend).  A three-level binary shifter on the dword number: selects, no
     * branches -- as a `switch` this is a dozen exec-mask regions. */
    __device__ __forceinline__ void windowBytes(uint32_t q, uint32_t &x0, uint32_t &x1) const
    {
        const uint32_t o = q - (wend - ENTRY);
        const bool b1 = (o & 4u) != 0, b2 = (o & 8u) != 0, b4 = (o & 16u) != 0;
        const uint32_t W1 = W.x, W2 = W.y, W3 = W.z, W4 = W.w;
        if (!kDeep) {
            const uint32_t T0 = b1 ? W1 : W0, T1 = b1 ? W2 : W1, T2 = b1 ? W3 : W2, T3 = b1 ? W4 : W3, T4 = b1 ? 0u : W4;
            const uint32_t U0 = b2 ? T2 : T0, U1 = b2 ? T3 : T1, U2 = b2 ? T4 : T2;
            const uint32_t lo = b4 ? W4 : U0;
            x0 = __builtin_amdgcn_alignbyte(U1, lo, o & 3u);
            x1 = __builtin_amdgcn_alignbyte(U2, U1, o & 3u);
        } else {                                           /* nine dwords W0..W8, dword number 0..8: shifts by 4, 2, 1 (8 = W8 alone) */
            const bool b8 = (o & 32u) != 0;
            const uint32_t W5 = V.x, W6 = V.y, W7 = V.z, W8 = V.w;
            const uint32_t E0 = b8 ? W8 : (b4 ? W4 : W0), E1 = b4 ? W5 : W1, E2 = b4 ? W6 : W2, E3 = b4 ? W7 : W3, E4 = b4 ? W8 : W4, E5 = W5;
            const uint32_t F0 = b2 ? E2 : E0, F1 = b2 ? E3 : E1, F2 = b2 ? E4 : E2, F3 = b2 ? E5 : E3;
            const uint32_t G0 = b1 ? F1 : F0, G1 = b1 ? F2 : F1, G2 = b1 ? F3 : F2;
            x0 = __builtin_amdgcn_alignbyte(G1, G0, o & 3u);
            x1 = __builtin_amdgcn_alignbyte(G2, G1, o & 3u);
        }
    }
    /* kDeep: input bytes q+8 .. q+23 out of the nine-dword window (a long slot's extension: the caller has checked that the
     * window holds all 24 bytes from q, so q - (wend - 36) <= 12: the dword number is 0..3 and only the shifts by 2 and 1 are needed) */
    __device__ __forceinline__ void windowBytesExt(uint32_t q, uint32_t &y0, uint32_t &y1, uint32_t &y2, uint32_t &y3) const
    {
        const uint32_t o = q - (wend - ENTRY);
        const bool b1 = (o & 4u) != 0, b2 = (o & 8u) != 0;
        const uint32_t W1 = W.x, W2 = W.y, W3 = W.z, W4 = W.w, W5 = V.x, W6 = V.y, W7 = V.z, W8 = V.w;
        (void)W1;
        /* dwords 2 .. 6 behind dword number d = o >> 2 (0..3): d + 2 .. d + 6 <= 9; W9 does not exist and is never needed (o <= 12 means
         * d = 3 only with o = 12: the 24 bytes end with W8) */
        const uint32_t F2 = b2 ? W4 : W2, F3 = b2 ? W5 : W3, F4 = b2 ? W6 : W4, F5 = b2 ? W7 : W5, F6 = b2 ? W8 : W6, F7 = b2 ? W8 : W7;
        const uint32_t G2 = b1 ? F3 : F2, G3 = b1 ? F4 : F3, G4 = b1 ? F5 : F4, G5 = b1 ? F6 : F5, G6 = b1 ? F7 : F6;
        y0 = __builtin_amdgcn_alignbyte(G3, G2, o & 3u);
        y1 = __builtin_amdgcn_alignbyte(G4, G3, o & 3u);
        y2 = __builtin_amdgcn_alignbyte(G5, G4, o & 3u);
        y3 = __builtin_amdgcn_alignbyte(G6, G5, o & 3u);
    }

    /* A walk starts in the JUMP table (tables.cpp): the queue entry {position, 20 input bytes} is its first window,
     * and the prefilter has just found its first four bytes to be -- probably -- a pattern prefix, so the slot at
     * hash(those four bytes) takes it four or more bytes deep with its first gathered load (ks = 0: the bucket is
     * the slot itself).  If the slot is somebody else's, consume() restarts the walk in the initial state's bucket. */
    __device__ __forceinline__ void start(const Ctx &c, const u32x4 &ea, const uint32_t eb0, const uint32_t eb1, const u32x4 &ec, const uint32_t *shortBits)
    {
        pos = ea.x; match = 0; depth = 0; b0 = ea.y & 0xFF;
        W0 = ea.y; W.x = ea.z; W.y = ea.w; W.z = eb0; W.w = eb1;
        if (kDeep) V = ec;
        wend = pos + ENTRY;
        needWin = false; needSlot = true; longWalk = false; haveE = false; needExt = false;
        /* a pattern of one or two bytes matches here (shortBits: the exact 2-byte bitmap, only given when the set has
         * such patterns): the prefix passes a final state, so it has no jump slot -- straight to the initial state's
         * bucket instead of finding that out a round later */
        const bool viaRoot = shortBits != nullptr && testBit(shortBits, ea.y & 0xFFFFu) != 0;
        first = !viaRoot;
        row = viaRoot ? c.rootRow : c.jumpBase + ((ea.y * pfac::kJumpMul) >> c.jumpShift);
        ks = viaRoot ? kRootKs : 0u;
    }
    __device__ __forceinline__ u32x4 loadSlot(const Ctx &c, uint32_t idx) const
    {
        if (TEX) return __builtin_amdgcn_raw_buffer_load_b128(c.rsrc, (int)(idx * 16u), 0, PFAC_SLOT_AUX);
        return c.slots[idx];
    }
    /* spec (wave-uniform, kSpec only): the wave expects long slots -- its stream has been full of near misses -- and fetches
     * the extension unit of a wide bucket's slot WITH the header.  Otherwise a unit is fetched when a long slot's header bytes
     * have matched (one more trip of the scan loop for that walk, like a window that has to be re-fetched): on benign input
     * that is rare, and a unit fetched for nothing is a gathered load of a cold line. */
    __device__ __forceinline__ void issue(const Ctx &c, bool spec)
    {
        if (needSlot) {
#if defined(PFAC_EXP_CONFINE)           /* timing experiment: every slot load inside one window of the table; results are wrong */
            const uint32_t idx = (row + chainHashSlot(ks, b0)) & (PFAC_EXP_CONFINE - 1u);
#else
            const uint32_t idx = row + chainHashSlot(ks, b0);
#endif
            t = loadSlot(c, idx);
            if (kSpec) {
                haveE = spec & ((ks & pfac::kSlotWide) != 0);
                if (haveE) E = loadSlot(c, idx + c.extDelta);
            }
        } else if (kSpec && needExt) {                         /* the header in t is a long slot whose unit did not come with it */
            E = loadSlot(c, row + chainHashSlot(ks, b0) + c.extDelta);
            haveE = true;
        }
        if (needWin) {                                         /* rare: the walk is deeper than its entry */
            W = loadWindow16(c.in32, pos + depth + 1);                 /* pos + depth = position of the edge byte b0 */
            if (kDeep) V = loadWindow16(c.in32, pos + depth + 17);
            wend = ((pos + depth + 1) & ~3u) + (kDeep ? 32u : 16u);
        }
    }
    /* Finish the transition whose slot is in t (pfac::ChainSlot) on edge byte b0: compares the chain with the input behind
     * the edge byte, lands in the slot's end state and picks the next edge byte.  Straight-line but for the extension of a
     * long slot, which only runs when some lane of the wave has one; `match` is always valid.  False = the walk is over
     * (trap, or no successor). */
    __device__ __forceinline__ bool consume(const Ctx &c, bool spec)
    {
        const uint32_t q = pos + depth + 1;                    /* first byte behind the edge byte */
        const uint32_t meta = t.x;
        const uint32_t len = slotLen(meta);                    /* <= kChainMax, or <= kChainMaxWide in a wide bucket */
        const bool mine = (meta & (pfac::kSlotEmpty | 0xFFu)) == b0;
        const bool isLong = len > (uint32_t)pfac::kChainMax;
        const uint32_t lenIn = isLong ? (uint32_t)pfac::kChainMax : len;
        /* bytes of the window the header needs behind the edge byte: a short slot's chain and the next edge byte, a long
         * slot's eight header bytes */
        const bool coveredIn = q + lenIn + 1u <= wend;
        uint32_t x0, x1;
        windowBytes(q, x0, x1);
        const uint64_t diff = ((uint64_t)(x1 ^ t.w) << 32) | (x0 ^ t.z);
        /* the slot is this byte's (not empty, not another byte's), and the first lenIn chain bytes equal the
         * input: two shifts by less than 64 each, so that len == 0 shifts everything out */
        bool ok = mine & (((diff << 8) << (56u - 8u * lenIn)) == 0);
        /* a matching slot whose bytes are not all in the window: fetch them and come back.  A long slot whose header bytes
         * match (kSpec) needs all 24 bytes in the window (windowBytesExt shifts by at most three dwords) and its unit */
        tookLong = false;
        const bool longGo = kSpec && (isLong & ok & coveredIn);
        needWin = mine & (!coveredIn | (longGo & (q + 24u > wend)));
        needExt = longGo & !haveE;
        usedE = kSpec && (haveE & mine & isLong);
        const bool retry = needWin | needExt;
        needSlot = !retry;
        longWalk |= needWin;
        bool cont = true;
        if (retry) ok = true;
        if (!retry) {
            uint32_t next = (uint32_t)((((uint64_t)x1 << 32) | x0) >> (8u * lenIn)) & 0xFFu;   /* byte len (<= 7) behind the edge byte */
            if (__ballot(ok & isLong) != 0) {
                /* long slots (wide buckets): header byte 7, then chain bytes 8 .. len-1 in the extension unit against the
                 * input bytes 8 .. 23 behind the edge byte; the next edge byte is one of those */
                uint32_t y0, y1, y2, y3;
                u32x4 e;
                if (kSpec) {
                    windowBytesExt(q, y0, y1, y2, y3);
                    e = E;
                } else {
                    /* fetched now and waited for on the spot: rare where this path is compiled in (the compacted-output kernel,
                     * whose 20-byte window could not hold the bytes anyway) */
                    asm volatile("; pfac_ext_sync" ::: "memory");
                    const uint32_t at = row + chainHashSlot(ks, b0) + c.extDelta;
                    e = u32x4{0, 0, 0, 0};
                    u32x4 in4 = {0, 0, 0, 0};
                    uint32_t in1 = 0;
                    if (ok & isLong) {
                        e = loadSlot(c, at);
                        in4 = loadWindow16(c.in32, q + 8u);
                        in1 = c.in32[((q + 8u) >> 2) + 4u];
                    }
                    const uint32_t sh = (q + 8u) & 3u;
                    y0 = __builtin_amdgcn_alignbyte(in4.y, in4.x, sh);
                    y1 = __builtin_amdgcn_alignbyte(in4.z, in4.y, sh);
                    y2 = __builtin_amdgcn_alignbyte(in4.w, in4.z, sh);
                    y3 = __builtin_amdgcn_alignbyte(in1, in4.w, sh);
                    longWalk |= ok & isLong;                   /* the window is behind the walk now */
                    tookLong = ok & isLong;
                }
                const bool okLong = (((x1 ^ t.w) >> 24) == 0) & extensionEqual(e, y0, y1, y2, y3, len);
                ok &= !isLong | okLong;
                next = isLong ? byteOf16(y0, y1, y2, y3, len - 8u) : next;
            }
            const bool leaf = (meta & pfac::kSlotKMask) == 0;
            const int id = (int)(leaf ? t.y : t.w);                /* kSlotFinal: see pfac::ChainSlot (a final state with successors never ends a long slot) */
            match = (ok & ((meta & pfac::kSlotFinal) != 0)) ? id : match;   /* skipped chain states are never final */
            row = t.y;
            ks = meta;
            depth += 1 + len;
            b0 = next;
            cont = ok & !leaf;
        }
        /* the jump table does not know these four bytes (a collision, a final state on the way, a false positive
         * of the prefilter): the walk starts over in the initial state's bucket, one byte at a time */
        const bool restart = first & !ok;
        row = restart ? c.rootRow : row;
        ks = restart ? kRootKs : ks;
        depth = restart ? 0u : depth;
        b0 = restart ? (W0 & 0xFFu) : b0;
        cont |= restart;
        first = false;
        /* long walks (adversarial input): no more retry rounds -- a new window with every step, or, with the wide window,
         * whenever fewer bytes than the next step can consume are left of it (nine; 25 if the wave expects long slots and the
         * next bucket is wide: the slot, its unit and the window then come back together) */
        if (kDeep) {
            const bool wideNext = kSpec && (spec & ((ks & pfac::kSlotWide) != 0));
            if (!retry) needWin |= (longWalk | wideNext) & (wend < pos + depth + (wideNext ? 25u : 10u));
        } else {
            needWin |= longWalk;
        }
        return cont;
    }
};

/*
 * StageLane -- the walker of the full-result kernel (round 5).  Same split-phase protocol and the same transitions as
 * ChainLane, but the INPUT of a walk is read IN PLACE from the wave's LDS: a step reads the 12 (long slot: 28) bytes behind
 * its edge byte with three (seven) aligned LDS reads.  That replaces the nine-dword register window with its
 * 15-to-35-instruction shifter and every gathered window load.  Where in LDS depends on what the wave's stream looks like
 * (the kernel switches at moments when no walk is under way):
 *   TEXT mode   a queue entry is the candidate's offset in the staged chunk; when a lane takes it -- in the next trip of the
 *               loop, before the next chunk is staged -- the candidate's first kWalkEntryBytes input bytes are copied to the
 *               lane's own 32 bytes of LDS, and the walk reads those.  Text has few walks per chunk, all shallow, and a wave
 *               filters a chunk per trip of its loop.
 *   STAGE mode  (near-miss streams: BASELINE config 5) the wave keeps the last TWO chunks it filtered staged, each with the
 *               kWalkHalo bytes behind it, an entry is just {buffer, offset}, a walk reads the stage however deep it goes; a
 *               buffer is overwritten when no walk reads it any more.  There a chunk has dozens of walks 30 to 60 bytes deep:
 *               on that stream a third of a walk's gathered loads were window loads, and gathered loads (the CU's address
 *               path) and instruction issue are what the launch is bound by (profiles/r05_experiments.md).
 * A walk that runs off its entry / its stage loads input from global memory, waited for on the spot (text mode: deeper than
 * 19 bytes; stage mode: more than kWalkHalo bytes behind its chunk, i.e. patterns longer than ~100 bytes).
 */
/* The full-result kernel exists with BOTH walkers (template parameter STAGE): the register-window walker of rounds 2-4
 * (ChainLane: on text it is 1 % faster -- fewer scalar instructions and branches per trip of the loop, no copy when a walk
 * starts) and this one (19 % faster on the near-miss stream).  The host picks per launch from what the handle's last launch
 * found (ScanArgs::hostHint, written by the last block out): waves that ended in stage mode / with speculation on. */
constexpr uint32_t kWalkHalo = 128;            /* bytes behind a chunk that are staged with it (full-result kernel) */
constexpr uint32_t kWalkStageBytes = (uint32_t)pfac::kChunkTiles * 1024u + kWalkHalo;
constexpr uint32_t kWalkEntryBytes = 32;       /* text mode: input bytes a queue entry carries */
constexpr int kWalkReachShort = 13, kWalkReachLong = 29;   /* bytes from the edge byte on that a step reads (as whole dwords): 1 + 8 (+ 3), long slot: 1 + 24 (+ 3) */
struct StageView {                             /* the wave's two stage buffers (wave-uniform); in text mode the second one holds the entries' bytes */
    uint32_t addr[2];                          /* LDS byte address */
    uint32_t base[2];                          /* position of the staged chunk's first byte in this launch's input */
};
__device__ __forceinline__ uint32_t ldsWord(uint32_t byteAddr) { return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(byteAddr); }

template <bool TEX> struct StageLane {
    using Ctx = ChainCtx<TEX>;
    static constexpr bool kSpec = true;        /* wide buckets: header and extension unit can be fetched together (the wave decides: see the kernel) */
    uint32_t pos = 0;
    uint32_t sq = 0;                           /* LDS byte address of the edge byte */
    int rem = 0;                               /* bytes that can be read from LDS from the edge byte on */
    uint32_t delta = 0;                        /* position in the input = LDS address + delta */
    uint32_t row = 0;                          /* first slot of the current state's bucket */
    int match = 0;
    uint32_t ks = 0, b0 = 0;
    u32x4 t = {0, 0, 0, 0};
    u32x4 E;                                   /* the extension unit of the slot in flight (no initial value: see ChainLane) */
    uint32_t d0 = 0, d1 = 0, d2 = 0;           /* the three dwords that hold the 9 bytes behind the edge byte: read in issue(), like the slot, so that
                                                * consume() does not begin with an LDS round trip */
    bool inB = false;                          /* stage mode: which of the two stage buffers the walk reads */
    bool needSlot = true;
    bool haveE = false, needExt = false, usedE = false;
    bool first = false;                        /* the slot in flight comes from the jump table */
    bool ranOff = false;                       /* this step read its input from global memory */

    __device__ __forceinline__ u32x4 loadSlot(const Ctx &c, uint32_t idx) const
    {
        if (TEX) return __builtin_amdgcn_raw_buffer_load_b128(c.rsrc, (int)(idx * 16u), 0, PFAC_SLOT_AUX);
        return c.slots[idx];
    }
    /* longJump (wave-uniform): the wave expects long slots: the walk starts in the LONG jump table, whose slots fold up to 23
     * bytes behind the edge byte (the unit comes with the header: ks says "wide") */
    __device__ __forceinline__ void begin(const Ctx &c, uint32_t key, const uint32_t *shortBits, bool longJump)
    {
        match = 0;
        b0 = key & 0xFFu;
        needSlot = true; haveE = false; needExt = false;
        /* a pattern of one or two bytes matches here: the prefix passes a final state, so it has no jump slot (ChainLane::start) */
        const bool viaRoot = shortBits != nullptr && testBit(shortBits, key & 0xFFFFu) != 0;
        first = !viaRoot;
        row = viaRoot ? c.rootRow : (longJump ? c.jumpLongBase : c.jumpBase) + ((key * pfac::kJumpMul) >> c.jumpShift);
        ks = viaRoot ? kRootKs : (longJump ? pfac::kSlotWide : 0u);
    }
    /* stage mode: code = buffer << 31 | offset of the candidate in its chunk */
    __device__ __forceinline__ void startStage(const Ctx &c, const StageView &v, uint32_t code, const uint32_t *shortBits, bool longJump)
    {
        inB = (code >> 31) != 0;
        const uint32_t off = code & 0x7FFFFFFFu;
        sq = (inB ? v.addr[1] : v.addr[0]) + off;
        rem = (int)(kWalkStageBytes - off);
        pos = (inB ? v.base[1] : v.base[0]) + off;
        delta = pos - sq;
        const uint32_t a4 = sq & ~3u;
        begin(c, __builtin_amdgcn_alignbyte(ldsWord(a4 + 4u), ldsWord(a4), sq & 3u), shortBits, longJump);
    }
    /* text mode: off = offset of the candidate in the chunk staged at v.addr[0]; its first kWalkEntryBytes bytes are copied to
     * the lane's own place `mine` (dword aligned), where the walk reads them however long the chunk stays staged */
    __device__ __forceinline__ void startText(const Ctx &c, const StageView &v, uint32_t off, uint32_t mine, const uint32_t *shortBits, bool longJump)
    {
        inB = false;
        const uint32_t src = v.addr[0] + off, a4 = src & ~3u, sh = src & 3u;
        uint32_t d[9];
#pragma unroll
        for (int k = 0; k < 9; k++) d[k] = ldsWord(a4 + 4u * (uint32_t)k);
        u32x4 lo, hi;
        lo.x = __builtin_amdgcn_alignbyte(d[1], d[0], sh); lo.y = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
        lo.z = __builtin_amdgcn_alignbyte(d[3], d[2], sh); lo.w = __builtin_amdgcn_alignbyte(d[4], d[3], sh);
        hi.x = __builtin_amdgcn_alignbyte(d[5], d[4], sh); hi.y = __builtin_amdgcn_alignbyte(d[6], d[5], sh);
        hi.z = __builtin_amdgcn_alignbyte(d[7], d[6], sh); hi.w = __builtin_amdgcn_alignbyte(d[8], d[7], sh);
        *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(mine) = lo;
        *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(mine + 16u) = hi;
        sq = mine;
        rem = (int)kWalkEntryBytes;
        pos = v.base[0] + off;
        delta = pos - mine;
        begin(c, lo.x, shortBits, longJump);
    }
    __device__ __forceinline__ void issue(const Ctx &c, bool spec)
    {
        {
            const uint32_t a4 = (sq + 1u) & ~3u;
            d0 = ldsWord(a4); d1 = ldsWord(a4 + 4u); d2 = ldsWord(a4 + 8u);
        }
        if (needSlot) {
            const uint32_t idx = row + chainHashSlot(ks, b0);
            /* the top of the table (breadth first) is in LDS, as far as the block's LDS reaches -- every bucket of a small pattern
             * set --: such a header is read in consume(), with the input bytes (read here, into the registers a gathered load may
             * still be writing, it would have to wait for that load) */
            if (idx >= c.hotSlots) t = loadSlot(c, idx);
            haveE = spec & ((ks & pfac::kSlotWide) != 0);
            if (haveE) E = loadSlot(c, idx + c.extDelta);
        } else if (needExt) {                                  /* the header in t is a long slot whose unit did not come with it */
            E = loadSlot(c, row + chainHashSlot(ks, b0) + c.extDelta);
            haveE = true;
        }
    }
    /* 32 input bytes from position g on, from global memory, waited for on the spot: x0:x1 = bytes 0..7, y0..y3 = bytes 8..23 */
    __device__ __forceinline__ void loadDeep(const Ctx &c, uint32_t g, uint32_t &x0, uint32_t &x1, uint32_t &y0, uint32_t &y1, uint32_t &y2, uint32_t &y3) const
    {
        const u32x4 g0 = loadWindow16(c.in32, g), g1 = loadWindow16(c.in32, g + 16u);
        const uint32_t gs = g & 3u;
        x0 = __builtin_amdgcn_alignbyte(g0.y, g0.x, gs); x1 = __builtin_amdgcn_alignbyte(g0.z, g0.y, gs);
        y0 = __builtin_amdgcn_alignbyte(g0.w, g0.z, gs); y1 = __builtin_amdgcn_alignbyte(g1.x, g0.w, gs);
        y2 = __builtin_amdgcn_alignbyte(g1.y, g1.x, gs); y3 = __builtin_amdgcn_alignbyte(g1.z, g1.y, gs);
    }
    __device__ __forceinline__ bool consume(const Ctx &c)
    {
        const uint32_t a = sq + 1u;                            /* first byte behind the edge byte */
        const uint32_t a4 = a & ~3u, sh = a & 3u;
        if (c.hotSlots != 0) {   /* a header among the hot rows (issue() did not fetch it): row, ks and b0 still describe the bucket it is in */
            const uint32_t idx = row + chainHashSlot(ks, b0);
            if (idx < c.hotSlots) t = *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(c.hotAddr + idx * 16u);
        }
        const uint32_t meta = t.x;
        const uint32_t len = slotLen(meta);
        const bool mine = (meta & (pfac::kSlotEmpty | 0xFFu)) == b0;
        const bool isLong = len > (uint32_t)pfac::kChainMax;
        const uint32_t lenIn = isLong ? (uint32_t)pfac::kChainMax : len;
        uint32_t x0 = __builtin_amdgcn_alignbyte(d1, d0, sh), x1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
        uint32_t y0 = 0, y1 = 0, y2 = 0, y3 = 0;
        const bool deep = rem < kWalkReachShort;               /* the walk has run off its entry / its stage */
        ranOff = deep;
        if (__ballot(deep) != 0) {
            asm volatile("; pfac_deep_sync" ::: "memory");
            if (deep) loadDeep(c, a + delta, x0, x1, y0, y1, y2, y3);
        }
        const uint64_t diff = ((uint64_t)(x1 ^ t.w) << 32) | (x0 ^ t.z);
        bool ok = mine & (((diff << 8) << (56u - 8u * lenIn)) == 0);
        needExt = isLong & ok & !haveE;                        /* a long slot whose header bytes match, its unit not here: fetch it and come back */
        usedE = haveE & mine & isLong;
        needSlot = !needExt;
        bool cont = true;
        if (needExt) ok = true;
        if (!needExt) {
            uint32_t next = (uint32_t)((((uint64_t)x1 << 32) | x0) >> (8u * lenIn)) & 0xFFu;
            if (__ballot(ok & isLong) != 0) {
                /* long slots (wide buckets): header byte 7, then chain bytes 8 .. len-1 in the extension unit against the
                 * input bytes 8 .. 23 behind the edge byte; the next edge byte is one of those */
                const bool deepLong = (rem < kWalkReachLong) & !deep & ok & isLong;
                ranOff |= deepLong;
                if (__ballot(deepLong) != 0) {
                    asm volatile("; pfac_deep_sync" ::: "memory");
                    uint32_t u0, u1;
                    if (deepLong) loadDeep(c, a + delta, u0, u1, y0, y1, y2, y3);
                }
                const uint32_t d3 = ldsWord(a4 + 12u), d4 = ldsWord(a4 + 16u), d5 = ldsWord(a4 + 20u), d6 = ldsWord(a4 + 24u);
                const uint32_t f0 = __builtin_amdgcn_alignbyte(d3, d2, sh), f1 = __builtin_amdgcn_alignbyte(d4, d3, sh),
                               f2 = __builtin_amdgcn_alignbyte(d5, d4, sh), f3 = __builtin_amdgcn_alignbyte(d6, d5, sh);
                if (__ballot(ranOff) == 0) {                   /* the usual case: nobody's bytes came from global memory */
                    y0 = f0; y1 = f1; y2 = f2; y3 = f3;
                } else {
                    const bool fromLds = rem >= kWalkReachLong;
                    y0 = fromLds ? f0 : y0; y1 = fromLds ? f1 : y1; y2 = fromLds ? f2 : y2; y3 = fromLds ? f3 : y3;
                }
                const bool okLong = (((x1 ^ t.w) >> 24) == 0) & extensionEqual(E, y0, y1, y2, y3, len);
                ok &= !isLong | okLong;
                next = isLong ? byteOf16(y0, y1, y2, y3, len - 8u) : next;
            }
            const bool leaf = (meta & pfac::kSlotKMask) == 0;
            const int id = (int)(leaf ? t.y : t.w);
            match = (ok & ((meta & pfac::kSlotFinal) != 0)) ? id : match;
            /* the jump table does not know these four bytes: the walk starts over in the initial state's bucket, on the same edge byte */
            const bool restart = first & !ok;
            row = restart ? c.rootRow : t.y;
            ks = restart ? kRootKs : meta;
            sq = restart ? sq : sq + 1u + len;
            rem = restart ? rem : rem - (int)(1u + len);
            b0 = restart ? b0 : next;
            cont = restart | (ok & !leaf);
        }
        first = false;
        return cont;
    }
};

/* The longest pattern that starts at in[p], walked through the chained table from the initial state's bucket with every
 * read checked against `readable` (a pattern that would run past the input does not match: ref PFAC_CPU.cpp:60-100, the
 * walk stops at the last byte).  Same transition rule as ChainLane::advance, one byte compare at a time: for the few
 * thousand positions at the ends of an input. */
template <bool TEX>
__device__ int boundedWalk(const ChainCtx<TEX> &c, const unsigned char *in, size_t p, size_t readable)
{
    uint32_t row = c.rootRow, ks = kRootKs;
    size_t at = p;                                   /* position of the edge byte */
    int match = 0;
    while (at < readable) {
        const uint32_t b0 = in[at];
        const uint32_t idx = row + chainHashSlot(ks, b0);
        u32x4 t;
        if (TEX) t = __builtin_amdgcn_raw_buffer_load_b128(c.rsrc, (int)(idx * 16u), 0, 0);
        else t = c.slots[idx];
        if ((t.x & (pfac::kSlotEmpty | 0xFFu)) != b0) break;
        const uint32_t len = slotLen(t.x);
        if (at + len >= readable) break;             /* the chain's bytes at+1 .. at+len must exist */
        const uint64_t chain = ((uint64_t)t.w << 32) | t.z;
        bool ok = true;
        for (uint32_t k = 0; k < len && k < 8u; k++) ok &= in[at + 1 + k] == (uint32_t)((chain >> (8u * k)) & 0xFFu);
        if (ok && len > 8u) {                        /* a long slot of a wide bucket: chain bytes 8 .. len-1 in its extension unit */
            const uint32_t eidx = idx + c.extDelta;
            u32x4 e;
            if (TEX) e = __builtin_amdgcn_raw_buffer_load_b128(c.rsrc, (int)(eidx * 16u), 0, 0);
            else e = c.slots[eidx];
            const uint64_t lo = ((uint64_t)e.y << 32) | e.x, hi = ((uint64_t)e.w << 32) | e.z;
            for (uint32_t k = 8; k < len; k++) ok &= in[at + 1 + k] == (uint32_t)(((k < 16u ? lo : hi) >> (8u * (k & 7u))) & 0xFFu);
        }
        if (!ok) break;
        const bool leaf = (t.x & pfac::kSlotKMask) == 0;
        if (t.x & pfac::kSlotFinal) match = (int)(leaf ? t.y : t.w);
        if (leaf) break;
        row = t.y;
        ks = t.x;
        at += 1 + len;
    }
    return match;
}

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

/* inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts, the gfx9 wave scan) */
__device__ __forceinline__ uint32_t waveInclusiveScan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   /* row_shr:1 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   /* row_shr:2 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   /* row_shr:4 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   /* row_shr:8 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   /* row_bcast:15 -> rows 1, 3 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   /* row_bcast:31 -> rows 2, 3 */
    return v;
}

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
template <bool TEX, bool HAS_SHORT, bool REDUCE, int kWalkSets, bool STAGE>
__global__ __launch_bounds__(kBlockThreads, PFAC_MIN_WAVES_PER_SIMD) __attribute__((amdgpu_num_vgpr(kCompilerVgprs)))
void pfac_scan_filter(ScanArgs a)
{
    constexpr int kTilesPerIter = kGroupTiles;
    constexpr int kChunkBytes = kTilesPerIter * kTileBytes;    /* input bytes a wave stages at a time */
    using WCtx = ChainCtx<TEX>;
    constexpr uint32_t kEntry = REDUCE ? kEntryBytes : kEntryBytesFull;
    /* full-result kernel: walks read their input from the wave's two staged chunks (StageLane), a queue entry is {buffer, offset};
     * compacted-output kernel (16 scanning waves, no LDS to spare): the input travels with the entry and lives in registers */
    constexpr bool kStageWalk = !REDUCE && STAGE;
    using WLane = std::conditional_t<kStageWalk, StageLane<TEX>, ChainLane<TEX, kEntry>>;
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
    uint32_t *sStageAll = sQueueCAll + ((REDUCE || kStageWalk) ? 0 : kScanners * kQCap * 4);   /* per scanning wave: the chunk being filtered + the bytes behind it (kStageWalk: and the chunk before it) */
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
    uint16_t *list = reinterpret_cast<uint16_t *>(sListAll + wave * (kListCap / 2));
    const uint32_t n = (uint32_t)a.n;               /* < 2^32: the launcher splits larger inputs */
    const Lds lds{sGram3, sLadder, sFinal3, sShort,
                  35u - (uint32_t)a.log2Bits /* product -> byte address of the level-1 dword */, 32u - (uint32_t)a.log2BitsLad, 32u - (uint32_t)a.log2BitsF3};
    /* one probe of the prefix ladder: bit `v >> shiftLad` of the bitmap, in bit 0 of the result (the bits above it are garbage) */
    auto ladProbe = [&](uint32_t v) -> uint32_t {
        const uint32_t idx = v >> lds.shiftLad;
        return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(((idx >> 3) & ~3u) + kLadderLdsOffset) >> (idx & 31u);
    };
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
                        if (!REDUCE) ec = queueC[qi];
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
    asm volatile("v_mov_b32 %0, %1" : "=v"(vShift3) : "s"(REDUCE ? 0xFFFCu : lds.shift3));      /* compacted-output kernel: the address mask of gram1 */
    asm volatile("v_mov_b32 %0, %1" : "=v"(vGram3Mul) : "s"(pfac::kGram3Mul));
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
        walkConsume();
        PFAC_TICK(0);
        flushWalks = chunk == kEnd && listAt == listEnd;
        /* ---- 2. hand idle walker lanes new positions, start the next transition of
         *         every live walk */
        walkRefill();
        PFAC_TICK(1);
        walkIssue();
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
                            if (REDUCE) {
                                /* gram1: byte address of the dword = bits 18..31 of the product, times four = the product's high half
                                 * AND 0xFFFC -- one SDWA instruction (a shift and an AND otherwise) */
                                uint32_t addr;
                                asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(addr) : "v"(product), "v"(vShift3));
                                word[q] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(addr + kGram1LdsOffset);
                            } else {
                                word[q] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((product >> vShift3) & ~3u);
                            }
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
        if (listAt == listEnd && __ballot(hits != 0) != 0) {
            const uint32_t cnt = (uint32_t)__builtin_popcount(hits);
            const uint32_t incl = waveInclusiveScan(cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t idx = incl - cnt;
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
            while (hits != 0 && idx < kListCap) {       /* divergent: as many rounds as the busiest lane has hits */
                list[idx] = (uint16_t)(((uint32_t)lane << 5) | (uint32_t)__builtin_ctz(hits));
                idx++;
                hits &= hits - 1;
            }
            PFAC_TICK(10);
            const uint32_t listed = dense ? 0u : (total < kListCap ? total : kListCap);
            stHits += listed;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            PFAC_TICK(4);
            /* ---- 5a. every listed hit: its first four bytes against level 4 of the prefix ladder (are they a pattern
             * prefix at all?), the length-3 bitmap and the exact 2-byte bitmap.  Survivors stay in the list, compacted
             * in place (a wave's LDS accesses execute in order: every lane has read its code before any lane writes, and
             * the k-th round writes below the codes it has read); bit 15 = "walk, whatever follows" (an S node at depth
             * 4, or a pattern of up to three bytes matches here). */
            uint32_t kept = 0;
            for (uint32_t base = 0; base < listed; base += 64u) {
                const bool act = base + (uint32_t)lane < listed;
                const uint32_t code = act ? listCode(base) : 0u;
                const uint32_t o = ((code & 0x10u) << 6) | ((code >> 1) & 0x3F0u) | (code & 0xFu);       /* byte offset inside the chunk: tile, lane, position */
                const uint32_t at = o >> 2, sh = o & 3u;
                const uint32_t x = __builtin_amdgcn_alignbyte(stage[at + 1], stage[at], sh);
                const uint32_t h = x * pfac::kLadMul0;
                uint32_t sHit, gHit;
                if (REDUCE) {                                      /* every 4-byte pattern prefix walks: prefix4, two probes (LDS address 0) */
                    auto probe4 = [&](uint32_t v) -> uint32_t {
                        const uint32_t idx = v >> (32 - pfac::kPrefix4Log2);
                        return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((idx >> 3) & ~3u) >> (idx & 31u);
                    };
                    sHit = probe4(h) & probe4(h * pfac::kLadMulS) & 1u;
                    gHit = 0;
                } else {
                    sHit = ladProbe(h) & ladProbe(h * pfac::kLadMulS) & 1u;
                    gHit = ladProbe(h * pfac::kLadMulG) & ladProbe(h * pfac::kLadMulG2) & 1u;
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
            stCand += kept;
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
            const uint32_t room = kQCap - (qv - qh);
            const uint32_t take = room < want ? room : want;
            if (left == 0 || (take != want && take < kAppendMin)) break;
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
            if (skipLadder) { ladderSkip--; walk |= und; und = 0; }
            if (!REDUCE && __ballot(und != 0) != 0) {
                uint32_t hl[pfac::kLadderLevels];
                hl[0] = x0 * pfac::kLadMul0;
#pragma unroll
                for (int lv = 1; lv < pfac::kLadderLevels; lv++) {
                    const uint32_t xw = lv <= 2 ? x1 : lv <= 4 ? x2 : lv <= 6 ? x3 : x4;
                    hl[lv] = (hl[lv - 1] ^ ((lv & 1) ? (xw & 0xFFFFu) : (xw >> 16))) * pfac::kLadMul;
                }
#pragma unroll
                for (int lv = 1; lv < pfac::kLadderLevels; lv++) {
                    /* one early exit, in the middle: a check per level makes every level wait for the LDS reads of the one before
                     * it, and on text a batch almost always has a candidate that follows some long keyword to the last levels */
                    if (lv == 5 && __ballot(und != 0) == 0) break;
                    const uint32_t h = hl[lv];
                    const uint32_t sHit = ladProbe(h) & ladProbe(h * pfac::kLadMulS);      /* bit 0; und is 0 or 1 */
                    walk |= und & sHit;
                    if (lv == pfac::kLadderLevels - 1) und = 0;      /* the last level has S nodes only */
                    else und &= ladProbe(h * pfac::kLadMulG) & ~sHit;
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
                if (!REDUCE) {                                  /* bytes 20..35: read now, for the few that are kept */
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
                published[lane] = (unsigned long long)votes | ((unsigned long long)(kStageWalk ? 1u : 0u) << 32);
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

/* ------------------------------------------- reference-shaped kernel (PFACX_KERNEL_REFTABLE) */

/* One thread per input byte, no prefilter: the reference's algorithm with only the initial-state row
 * staged in LDS.  Alignment-agnostic, 64-bit positions.  Produces results for positions [0, owned);
 * walks may read up to a.n (owned <= n). */
constexpr size_t kChunkBytesDev = (size_t)pfac::kChunkTiles * kTileBytes;
template <int MODE>
__global__ __launch_bounds__(256) void pfac_scan_naive(ScanArgs a)
{
    __shared__ int sInit[pfac::kCharSet];
    if (a.owned == 0) return;
    sInit[threadIdx.x] = a.initialRow[threadIdx.x];
    __syncthreads();
    const Lookup<MODE> lookup(a);
    const size_t n = a.n;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < a.owned; j += stride) {
        int state = sInit[a.in[j]];
        int match = 0;
        if (state != kTrap) {
            if (state <= a.numFinal) match = state;
            for (size_t pos = j + 1; pos < n; pos++) {
                state = lookup(state, a.in[pos]);
                if (state == kTrap) break;
                if (state <= a.numFinal) match = state;
            }
        }
        if (a.reducePos == nullptr) {
            a.out[j] = match;
        } else if (match > 0) {                             /* compacted output: a pair behind the others, any order (orderPairs) */
            const unsigned int at = atomicAdd(a.reduceCount, 1u);
            a.out[at] = match;
            a.reducePos[at] = (int)(a.reduceBase + (unsigned int)j);
        }
    }
}

/* ---------------------------------------------------------- tiled kernel */

/*
 * pfac_scan_tiled -- one position per thread-slot, everything a walk touches first in LDS.  The kernel of small calls
 * (PFACX_KERNEL_AUTO below kSmallInput), of PFACX_KERNEL_NAIVE, and of the chunks the filter kernel found pattern-dense.
 * Shape of the reference's kernel (PFAC/src/PFAC_kernel.cu:377-458): tile + halo staged in shared memory with wide
 * coalesced loads (:405-417), positions walked out of LDS bytes (:255-299), results written as whole coalesced lines
 * (:443-457) -- on the chained table (one 16-byte slot per transition + single-successor chain, tables.cpp) instead of
 * one gathered table word per byte (dense) or two dependent loads (hashed, PFAC_kernel_spaceDriven.cu:76-124).
 *
 * A wave owns a GROUP of TILES x 1 KiB of input at a time; nothing is shared between waves after the block has filled
 * its LDS tables, so the kernel has no barrier in its loop:
 *   load     16 B per lane and tile (one 1 KiB-contiguous instruction each) + the 128 bytes behind the group (lanes
 *            0..31, a dword each); group and halo go to the wave's LDS stage.
 *   results  every result of the group is stored as zero straight away: 16 B per lane, 1 KiB contiguous per
 *            instruction, non-temporal -- whole lines, nothing read.  A walk that ends in a match overwrites its zero
 *            after an s_waitcnt vmcnt(0) (the zero is in L2 by then; same wave, same address: ordered).  With
 *            ScanArgs::reducePos nothing is zeroed and the matches are appended to the pair list instead (one atomic
 *            per wave and walk set).
 *   rows     the initial state's 256-wide transition row (the chained root bucket, 4 KiB) and the buckets the initial
 *            state's transitions land in, breadth first, as far as the CU's LDS reaches (ScanArgs::hotSlots), are in
 *            LDS for the whole launch; a walk leaves LDS only for a bucket behind them.
 *   early    before a position walks at all its first three bytes are looked up in the 3-gram bitmap (LDS; 1-2-byte
 *            patterns are folded into it, so the bytes behind the end of the input may be anything): a miss proves
 *            the result is 0 -- the dead state after at most three transitions, decided without taking them.
 *   walk     the survivors of all 64 lanes and all tiles of the group are compacted into one list (prefix sum of the
 *            lanes' hit counts), so a wave-wide step has up to 64 live walks however few positions survive; every lane
 *            runs WALKS independent walks at a time (the loads of a step are issued for all of them before the first is
 *            consumed: a step is a dependent round trip to LDS, L2 or beyond, and 4 KiB of text give ~200 survivors = one
 *            full round), and a step loop ends when __ballot says no lane of the wave is alive (PFAC_kernel.cu:299 is
 *            per thread).  Input bytes come from the stage; only a walk that runs more than 128 bytes past its group
 *            reads global memory, with every read checked against the end of the input.
 * Pointers may have any alignment: groups are cut at 16-byte aligned addresses and the positions in front of the first
 * input byte / behind the last owned one are masked; aligned 16-byte loads that contain a valid byte cannot fault, all
 * others are not issued.  64-bit positions.
 */
constexpr uint32_t kTiledTile = 1024;                  /* input bytes per load instruction of a wave */
constexpr uint32_t kTiledHalo = 128;                   /* bytes behind the group that are staged with it */
constexpr uint32_t kTiledList = 256;                   /* 16-bit codes of surviving positions per pass (a group with more takes another pass) */
#ifndef PFAC_TILED_WALKS
#define PFAC_TILED_WALKS 4
#endif
#ifndef PFAC_TILED_TILES
#define PFAC_TILED_TILES 4
#endif
constexpr int kTiledWalks = PFAC_TILED_WALKS;          /* independent walks per lane */
constexpr int kTiledTilesBig = PFAC_TILED_TILES;       /* tiles per group: launches with megabytes in front of them */
constexpr uint32_t kTiledFar = 0x40000000u;            /* "the input ends nowhere near this group" */
/* per wave: stage, list and -- the one-tile shape of small calls -- the tile's results: there a call is as long as its slowest wave, and
 * a patch that has to wait until the zeros are in L2 is on that path (4 KiB call 8.3 -> 7.2 us); the big shape hides the wait behind
 * fifteen other waves and spends the LDS on hot rows */
constexpr uint32_t kTiledPairs = 32;                   /* compacted output: (position, id) pairs a wave stages in LDS before it appends them with one atomic */
constexpr uint32_t tiledWaveLds(int tiles) { return (uint32_t)tiles * kTiledTile + kTiledHalo + kTiledList * 2 + kTiledPairs * 8 + (tiles == 1 ? kTiledTile * 4 : 0); }

/* REF >= 0 (a TableMode): the same frame -- 16-byte loads, group + halo and the initial state's row in LDS, 3-gram early-out, whole
 * zero lines, compacted survivors, __ballot loop exit -- over the REFERENCE-layout table of the perf mode instead of the chained
 * one: a walk takes one byte per step through Lookup<REF> (dense: one gathered word, PFAC_kernel.cu:291; hashed: two dependent
 * loads, PFAC_kernel_spaceDriven.cu:76-124).  This is what PFACX_KERNEL_REFTABLE launches: the byte-compared tables of the
 * reference walked the way its kernels walk them (PFAC_kernel.cu:377-458), the independent implementation every parity test runs
 * beside the product kernels. */
template <bool TEX, int WALKS, int TILES, bool HOTALL, int REF = -1>
__global__ __launch_bounds__(1024) void pfac_scan_tiled(ScanArgs a)
{
    constexpr bool kRef = REF >= 0;
    constexpr uint32_t kGroup = (uint32_t)TILES * kTiledTile, kStage = kGroup + kTiledHalo;
    static_assert(kGroup <= 4096, "a position's code is 12 bits of offset in 16");
    static_assert(kTiledList >= 64u * (uint32_t)TILES, "dense mode parks the lanes' hit masks in the list's place");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool listMode = a.denseList != nullptr;
    unsigned int listed = 0;
    if (listMode) listed = a.work[a.denseWord];
    if (a.owned == 0 && listed == 0) return;          /* behind a filter launch that listed no dense chunk: before anything is loaded */

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6), waves = blockDim.x >> 6;
    const int words3 = 1 << (a.log2Bits - 5);
    uint32_t *sGram3 = reinterpret_cast<uint32_t *>(smem);                     /* LDS address 0: level 1 addresses it by number */
    u32x4 *sRoot = reinterpret_cast<u32x4 *>(sGram3 + words3);
    u32x4 *sHot = sRoot + pfac::kCharSet;
    unsigned char *waveBase = reinterpret_cast<unsigned char *>(sHot + a.hotSlots) + wave * tiledWaveLds(TILES);
    uint32_t *stage = reinterpret_cast<uint32_t *>(waveBase);
    uint16_t *list = reinterpret_cast<uint16_t *>(waveBase + kStage);
    constexpr bool kLdsResults = TILES == 1;               /* a sparse group's results are assembled in LDS and stored once, as whole lines */
    uint32_t *pairPos = reinterpret_cast<uint32_t *>(waveBase + kStage + kTiledList * 2), *pairId = pairPos + kTiledPairs;
    int *res = reinterpret_cast<int *>(waveBase + kStage + kTiledList * 2 + kTiledPairs * 8);
    /* ScanArgs::reportDense: the block's dense groups, groups, waves that are through -- in the pair staging of the block's first wave */
    uint32_t *blockAcc = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(sHot + a.hotSlots) + kStage + kTiledList * 2);
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
    {
        const u32x4 *g3 = reinterpret_cast<const u32x4 *>(a.gram3);
        u32x4 *s3 = reinterpret_cast<u32x4 *>(sGram3);
        for (int i = tid; i < words3 / 4; i += (int)blockDim.x) s3[i] = g3[i];
#ifndef PFAC_NO_DENSE_REPORT
        if (tid < 4) blockAcc[tid] = 0;
#endif
        if constexpr (kRef) {
            for (int i = tid; i < pfac::kCharSet; i += (int)blockDim.x) reinterpret_cast<int *>(sRoot)[i] = a.initialRow[i];   /* ref: the initial state's row in shared memory, PFAC_kernel.cu:396-403 */
        } else {
            for (int i = tid; i < pfac::kCharSet; i += (int)blockDim.x) sRoot[i] = a.chainSlots[a.rootRow + (uint32_t)i];
            for (uint32_t i = (uint32_t)tid; i < a.hotSlots; i += blockDim.x) sHot[i] = a.chainSlots[i];
        }
    }
    __syncthreads();
    const int *sInit = reinterpret_cast<const int *>(sRoot);
    (void)sInit;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(a.chainSlots), 0, (int)a.chainBytes, 0x00020000);
    const uint32_t shift3 = 35u - (uint32_t)a.log2Bits;
    const uint32_t hot = a.hotSlots;
    const bool reduce = a.reducePos != nullptr;
    /* compacted output: matches are staged per wave and appended kTiledPairs at a time -- one device counter answers ~90 atomics per
     * microsecond, and the Snort-style stream has 583 K matches per GiB in 500 K different (walk set, round)s: an atomic each was
     * 5 ms per GiB on top of a 1.4 ms scan */
    uint32_t staged = 0;                                   /* wave-uniform */
    auto flushPairs = [&]() {
        if (staged == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned int at = 0;
        if (lane == 0) at = atomicAdd(a.reduceCount, staged);
        at = (unsigned int)__builtin_amdgcn_readfirstlane((int)at);
        if ((uint32_t)lane < staged) { a.out[at + lane] = (int)pairId[lane]; a.reducePos[at + lane] = (int)pairPos[lane]; }
        staged = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
#ifndef PFAC_TILED_STATS
#define PFAC_TILED_STATS 0                     /* measurement build: wave-wide step iterations, live lane-steps, walks, passes, dense groups */
#endif
    uint32_t tsIter = 0, tsLane = 0, tsWalks = 0, tsPasses = 0, tsDense = 0, tsGroups = 0;

    /* One group: g16 = its 16-byte aligned first byte; `span` bytes from there may be loaded (a multiple of 16: up to the
     * end of the 16-byte block that holds the last input byte); positions [lo, hi) of the group get a result, written to
     * outGroup[offset]; `limit` = group offset of the first byte behind the input (a pattern cannot reach it);
     * posBase = position of the group's first byte in the caller's stream (compacted output) */
    auto scanGroup = [&](const unsigned char *g16, uint64_t span, uint32_t lo, uint32_t hi, uint32_t limit, int *outGroup, uint32_t posBase) {
        const uint32_t span32 = span > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)span;
        const bool whole = lo == 0 && (hi & (kTiledTile - 1u)) == 0;      /* whole tiles: all of the group, or -- a dense chunk -- its first ones */
        const bool bounded = limit < kGroup + a.maxWalk + 16u;           /* wave-uniform: a walk of this group can come near the end of the input */
        u32x4 dt[TILES];
        uint32_t follow = 0;
#pragma unroll
        for (int t = 0; t < TILES; t++) {
            dt[t] = u32x4{0, 0, 0, 0};
            const uint32_t off = (uint32_t)t * kTiledTile + (uint32_t)lane * 16u;
            if (off < span32) dt[t] = *reinterpret_cast<const u32x4 *>(g16 + off);
        }
        if (lane < (int)(kTiledHalo / 4) && kGroup + (uint32_t)lane * 4u < span32) follow = *reinterpret_cast<const uint32_t *>(g16 + kGroup + lane * 4);
#pragma unroll
        for (int t = 0; t < TILES; t++) reinterpret_cast<u32x4 *>(stage)[t * 64 + lane] = dt[t];
        if (lane < (int)(kTiledHalo / 4)) stage[kGroup / 4 + lane] = follow;
        /* ---- early-out: the 3-gram bitmap, 16 positions per lane and tile (bytes 16 lane .. 16 lane + 15, + 2 of the next lane) */
        uint32_t hits[TILES];
#pragma unroll
        for (int t = 0; t < TILES; t++) {
            const uint32_t dw[4] = {dt[t].x, dt[t].y, dt[t].z, dt[t].w};
            uint32_t nxtLane = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dw[0], 0x130, 0xf, 0xf, false);   /* wave_shl:1 */
            /* behind lane 63: the first dword of the next tile (lane 0 has it), or of the halo */
            const uint32_t wrap = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t + 1 < TILES ? dt[t + 1 < TILES ? t + 1 : t].x : follow));
            if (lane == 63) nxtLane = wrap;
            uint32_t h = 0;
#pragma unroll
            for (int b0 = 0; b0 < 16; b0 += 8) {
                uint32_t word[8], xs[9];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int j = (b0 + q) >> 2, i = (b0 + q) & 3;
                    const uint32_t nx = j < 3 ? dw[(j + 1) & 3] : nxtLane;
                    const uint32_t x = i == 0 ? dw[j] : i == 1 ? dw[j] >> 8 : __builtin_amdgcn_alignbyte(nx, dw[j], i);
                    const uint32_t product = (uint32_t)__umul24(x, pfac::kGram3Mul);
                    word[q] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((product >> shift3) & ~3u);
                    xs[q] = x;
                }
                xs[8] = b0 + 8 < 16 ? dw[(b0 + 8) >> 2] : nxtLane;
#pragma unroll
                for (int q = 0; q < 8; q++)
                    h = __builtin_amdgcn_alignbit((word[q] >> (xs[q] & 31u)) & (word[q] >> (xs[q + 1] & 31u)), h, 1);
            }
            h >>= 16;                                              /* bit i: position 1024 t + 16 lane + i */
            if (whole) {
                if ((uint32_t)t * kTiledTile >= hi) h = 0;
            } else {                                               /* a group at an end of the input: only positions [lo, hi) */
                const int at = t * (int)kTiledTile + lane * 16;
                const int first = (int)lo - at, last = (int)hi - at;
                const uint32_t f = first < 0 ? 0u : first > 16 ? 16u : (uint32_t)first, l = last < 0 ? 0u : last > 16 ? 16u : (uint32_t)last;
                h &= ((1u << l) - 1u) & ~((1u << f) - 1u);
            }
            hits[t] = h;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        /* bytes q .. q+7 of the group (q = offset from g16): from the stage, or -- a walk more than 128 bytes behind its
         * group -- from global memory, loading only dwords of 16-byte blocks that hold input */
        auto fetch = [&](uint32_t q, uint32_t &w0, uint32_t &w1) {
            uint32_t e0, e1, e2;
            if (q + 12u <= kStage) {
                const uint32_t *p = stage + (q >> 2);
                e0 = p[0]; e1 = p[1]; e2 = p[2];
            } else {
                const uint32_t at = q & ~3u;
                e0 = at < span32 ? *reinterpret_cast<const uint32_t *>(g16 + at) : 0u;
                e1 = at + 4u < span32 ? *reinterpret_cast<const uint32_t *>(g16 + at + 4u) : 0u;
                e2 = at + 8u < span32 ? *reinterpret_cast<const uint32_t *>(g16 + at + 8u) : 0u;
            }
            w0 = __builtin_amdgcn_alignbyte(e1, e0, q & 3u);
            w1 = __builtin_amdgcn_alignbyte(e2, e1, q & 3u);
        };

        /* bytes q .. q+15 of the group (the input behind a long slot's header) */
        auto fetch16 = [&](uint32_t q, uint32_t &y0, uint32_t &y1, uint32_t &y2, uint32_t &y3) {
            uint32_t e0, e1, e2, e3, e4;
            if (q + 20u <= kStage) {
                const uint32_t *p = stage + (q >> 2);
                e0 = p[0]; e1 = p[1]; e2 = p[2]; e3 = p[3]; e4 = p[4];
            } else {
                const uint32_t at = q & ~3u;
                e0 = at < span32 ? *reinterpret_cast<const uint32_t *>(g16 + at) : 0u;
                e1 = at + 4u < span32 ? *reinterpret_cast<const uint32_t *>(g16 + at + 4u) : 0u;
                e2 = at + 8u < span32 ? *reinterpret_cast<const uint32_t *>(g16 + at + 8u) : 0u;
                e3 = at + 12u < span32 ? *reinterpret_cast<const uint32_t *>(g16 + at + 12u) : 0u;
                e4 = at + 16u < span32 ? *reinterpret_cast<const uint32_t *>(g16 + at + 16u) : 0u;
            }
            y0 = __builtin_amdgcn_alignbyte(e1, e0, q & 3u);
            y1 = __builtin_amdgcn_alignbyte(e2, e1, q & 3u);
            y2 = __builtin_amdgcn_alignbyte(e3, e2, q & 3u);
            y3 = __builtin_amdgcn_alignbyte(e4, e3, q & 3u);
        };

        /* WALKS walks per lane from the group offsets o[] (alive[]: the lane has one), to the end: match[] = result */
        /* REF: one byte per step through the reference-layout table (byte q of the group from the stage, or -- beyond the halo -- from
         * global memory; a byte at or behind `limit` does not exist).  The lookups of all WALKS walks are issued before the first is used. */
        auto byteAt = [&](uint32_t q) -> uint32_t {
            uint32_t w;
            if (q < kStage) w = stage[q >> 2];
            else w = (q & ~3u) < span32 ? *reinterpret_cast<const uint32_t *>(g16 + (q & ~3u)) : 0u;
            return (w >> (8u * (q & 3u))) & 0xFFu;
        };
        auto runWalksRef = [&](const uint32_t (&o)[WALKS], bool (&alive)[WALKS], int (&match)[WALKS]) {
            if constexpr (kRef) {
                const Lookup<kRef ? REF : 0> lookup(a);
                uint32_t q[WALKS];
                int state[WALKS];
#pragma unroll
                for (int k = 0; k < WALKS; k++) {
                    q[k] = o[k]; match[k] = 0; state[k] = kTrap;
                    if (alive[k]) state[k] = sInit[byteAt(q[k])];
                    alive[k] = alive[k] & (state[k] != kTrap);
                    match[k] = (alive[k] && state[k] <= a.numFinal) ? state[k] : 0;
                    q[k]++;
                }
                for (;;) {
                    bool any = false;
#pragma unroll
                    for (int k = 0; k < WALKS; k++) { alive[k] = alive[k] & (q[k] < limit); any |= alive[k]; }
                    if (__ballot(any) == 0) break;             /* every walk of the wave is in the trap state (ref: per thread, PFAC_kernel.cu:299) */
                    int next[WALKS];
#pragma unroll
                    for (int k = 0; k < WALKS; k++) {
                        next[k] = kTrap;
                        if (alive[k]) next[k] = lookup(state[k], (int)byteAt(q[k]));
                    }
#pragma unroll
                    for (int k = 0; k < WALKS; k++) {
                        alive[k] = alive[k] & (next[k] != kTrap);
                        match[k] = (alive[k] && next[k] <= a.numFinal) ? next[k] : match[k];
                        state[k] = next[k];
                        q[k]++;
                    }
                }
            }
        };
        auto runWalksChained = [&](const uint32_t (&o)[WALKS], bool (&alive)[WALKS], int (&match)[WALKS]) {
            uint32_t q[WALKS], row[WALKS], ks[WALKS];
            /* one transition through slot s on the edge byte at q, w0:w1 = bytes q .. q+7 (ChainLane::advance, with the
             * end of the input checked: edge byte and chain must lie in front of `limit`) */
            auto step = [&](int k, const u32x4 &s, uint32_t w0, uint32_t w1) {
                const uint32_t meta = s.x, len = slotLen(meta);
                bool ok = alive[k] & ((meta & (pfac::kSlotEmpty | 0xFFu)) == (w0 & 0xFFu));
                if (bounded) ok &= q[k] + len < limit;
                if (__ballot(ok & (len != 0)) != 0) {              /* the top of a trie branches at every byte: no chain, nothing to compare */
                    const uint32_t x0 = __builtin_amdgcn_alignbyte(w1, w0, 1), x1 = w1 >> 8;
                    const uint64_t diff = ((uint64_t)(x1 ^ s.w) << 32) | (x0 ^ s.z);
                    const uint32_t lenIn = len < (uint32_t)pfac::kChainMax ? len : (uint32_t)pfac::kChainMax;
                    ok &= ((diff << 8) << (56u - 8u * lenIn)) == 0;
                    const bool isLong = len > (uint32_t)pfac::kChainMax;
                    if (__ballot(ok & isLong) != 0) {
                        /* a long slot of a wide bucket (pfac_context.h): header byte 7 and the chain bytes 8 .. len-1 of its
                         * extension unit against the 16 bytes from q + 8.  (row, ks and the edge byte still describe the bucket
                         * the slot came from.) */
                        const uint32_t ea = row[k] + chainHashSlot(ks[k], w0 & 0xFFu) + a.extDelta;
                        u32x4 e = {0, 0, 0, 0};
                        uint32_t y0 = 0, y1 = 0, y2 = 0, y3 = 0;
                        if (ok & isLong) {
                            if (TEX) e = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ea * 16u), 0, 0);      /* the units are not among the hot rows */
                            else e = a.chainSlots[ea];
                            fetch16(q[k] + 8u, y0, y1, y2, y3);
                        }
                        const uint32_t c0 = __builtin_amdgcn_alignbyte(e.x, s.w, 3), c1 = __builtin_amdgcn_alignbyte(e.y, e.x, 3),
                                       c2 = __builtin_amdgcn_alignbyte(e.z, e.y, 3), c3 = __builtin_amdgcn_alignbyte(e.w, e.z, 3);
                        const uint32_t n = len - 7u;                   /* 1..16 bytes from header byte 7 on */
                        const uint64_t lo = ((uint64_t)(y1 ^ c1) << 32) | (y0 ^ c0), hi = ((uint64_t)(y3 ^ c3) << 32) | (y2 ^ c2);
                        ok &= !isLong || (lowBytesZero(lo, n < 8u ? n : 8u) && lowBytesZero(hi, n > 8u ? n - 8u : 0u));
                    }
                }
                const bool leaf = (meta & pfac::kSlotKMask) == 0;
                const int id = (int)(leaf ? s.y : s.w);
                match[k] = (ok & ((meta & pfac::kSlotFinal) != 0)) ? id : match[k];
                row[k] = s.y;
                ks[k] = meta;
                q[k] += 1u + len;
                alive[k] = ok & !leaf;
            };
            {   /* first transition: the initial state's row, in LDS, indexed by the byte itself */
                u32x4 s[WALKS];
                uint32_t w0[WALKS], w1[WALKS];
#if PFAC_TILED_STATS
                tsIter++;
#pragma unroll
                for (int k = 0; k < WALKS; k++) { const uint32_t c = (uint32_t)__popcll(__ballot(alive[k])); tsLane += c; tsWalks += c; }
#endif
#pragma unroll
                for (int k = 0; k < WALKS; k++) {
                    q[k] = o[k]; match[k] = 0; row[k] = 0; ks[k] = 0;
                    w0[k] = w1[k] = 0; s[k] = u32x4{pfac::kSlotEmpty, 0, 0, 0};
                    if (alive[k]) { fetch(q[k], w0[k], w1[k]); s[k] = sRoot[w0[k] & 0xFFu]; }
                }
#pragma unroll
                for (int k = 0; k < WALKS; k++)
                    if (__ballot(alive[k]) != 0) step(k, s[k], w0[k], w1[k]);      /* a walk set nobody is in costs a branch */
            }
            for (;;) {
                bool any = false;
#pragma unroll
                for (int k = 0; k < WALKS; k++) any |= alive[k];
                if (__ballot(any) == 0) break;                 /* every walk of the wave is in the dead state (or matched at a leaf) */
#if PFAC_TILED_STATS
                tsIter++;
#pragma unroll
                for (int k = 0; k < WALKS; k++) tsLane += (uint32_t)__popcll(__ballot(alive[k]));
#endif
                u32x4 s[WALKS];
                uint32_t w0[WALKS], w1[WALKS];
#pragma unroll
                for (int k = 0; k < WALKS; k++) {
                    w0[k] = w1[k] = 0; s[k] = u32x4{pfac::kSlotEmpty, 0, 0, 0};
                    if (alive[k]) {
                        fetch(q[k], w0[k], w1[k]);
                        const uint32_t at = row[k] + chainHashSlot(ks[k], w0[k] & 0xFFu);
                        if (HOTALL || at < hot) s[k] = sHot[at];
                        else if (TEX) s[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(at * 16u), 0, 0);
                        else s[k] = a.chainSlots[at];
                    }
                }
#pragma unroll
                for (int k = 0; k < WALKS; k++)
                    if (__ballot(alive[k]) != 0) step(k, s[k], w0[k], w1[k]);
            }
        };
        auto runWalks = [&](const uint32_t (&o)[WALKS], bool (&alive)[WALKS], int (&match)[WALKS]) {
            if constexpr (kRef) runWalksRef(o, alive, match);
            else runWalksChained(o, alive, match);
        };
        /* compacted output: the matches of a walk set join the wave's staged pairs */
        auto appendPairs = [&](const uint32_t (&o)[WALKS], const int (&match)[WALKS]) {
#pragma unroll
            for (int k = 0; k < WALKS; k++) {
                const bool has = match[k] != 0;
                const uint64_t m = __ballot(has);
                if (m) {
                    const uint32_t n = (uint32_t)__popcll(m);
                    if (staged + n > kTiledPairs) flushPairs();
                    if (n > kTiledPairs) {                         /* match-dense input: this set alone is worth an atomic */
                        unsigned int at = 0;
                        if (lane == 0) at = atomicAdd(a.reduceCount, n);
                        at = (unsigned int)__builtin_amdgcn_readfirstlane((int)at) + laneRankIn(m);
                        if (has) { a.out[at] = match[k]; a.reducePos[at] = (int)(posBase + o[k]); }
                    } else {
                        const uint32_t at = staged + laneRankIn(m);
                        if (has) { pairPos[at] = posBase + o[k]; pairId[at] = (uint32_t)match[k]; }
                        staged += n;
                    }
                }
            }
        };

        uint32_t cnt = 0;
#pragma unroll
        for (int t = 0; t < TILES; t++) cnt += (uint32_t)__builtin_popcount(hits[t]);
        const uint32_t survivors = (uint32_t)__builtin_amdgcn_readlane((int)waveInclusiveScan(cnt), 63);
        tsGroups++;
        if (survivors * 2u >= hi - lo) tsDense++;
        if (survivors * 2u >= hi - lo) {
            /* ---- DENSE group (half of its positions or more survive: short patterns over text, runs of a pattern byte):
             * compaction would cost more than idle lanes.  Position p = 256 r + 64 k + lane walks in round r, walk k: the
             * lanes of a walk are 64 consecutive positions -- their stage bytes are 16 consecutive dwords, their results one
             * 256-byte store, and nothing is zeroed first.  The hit masks go through LDS (the list's place). */
#pragma unroll
            for (int t = 0; t < TILES; t++) list[t * 64 + lane] = (uint16_t)hits[t];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t base = lo & ~255u; base < hi; base += 64u * (uint32_t)WALKS) {
                uint32_t o[WALKS];
                int match[WALKS];
                bool alive[WALKS], mine[WALKS];
#pragma unroll
                for (int k = 0; k < WALKS; k++) {
                    o[k] = base + (uint32_t)k * 64u + (uint32_t)lane;
                    mine[k] = o[k] >= lo && o[k] < hi;
                    alive[k] = mine[k] && (((uint32_t)list[o[k] >> 4] >> (o[k] & 15u)) & 1u) != 0;
                }
                runWalks(o, alive, match);
                if (reduce) appendPairs(o, match);
                else {
#pragma unroll
                    for (int k = 0; k < WALKS; k++)
                        if (mine[k]) __builtin_nontemporal_store(match[k], outGroup + o[k]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            return;
        }
        /* ---- SPARSE group: every result is stored as zero now -- 16 B per lane, 1 KiB contiguous per instruction, whole
         * lines, nothing read; they are 80 % of the call's traffic -- and the few walks that end in a match overwrite theirs */
        if (!reduce) {
            const i32x4 zero = {0, 0, 0, 0};
            if (kLdsResults) {
#pragma unroll
                for (int k = 0; k < 4 * TILES; k++) reinterpret_cast<i32x4 *>(res)[k * 64 + lane] = zero;
            } else if (whole) {
#pragma unroll
                for (int k = 0; k < 4 * TILES; k++)
                    if ((uint32_t)k * 256u < hi) __builtin_nontemporal_store(zero, reinterpret_cast<i32x4 *>(outGroup) + k * 64 + lane);
            } else {
                for (uint32_t p = lo + (uint32_t)lane; p < hi; p += 64u) outGroup[p] = 0;
            }
        }
        /* the survivors, compacted: passes of up to kTiledList positions, each walked 64 x WALKS at a time.  A group with
         * more survivors than one pass takes lists at most kTiledList / 64 of every lane per pass: all lanes emit for a few
         * trips, instead of the first few lanes for as many trips as they have hits. */
        const bool crowded = survivors > kTiledList;           /* wave-uniform */
        for (;;) {
            cnt = 0;
#pragma unroll
            for (int t = 0; t < TILES; t++) cnt += (uint32_t)__builtin_popcount(hits[t]);
            if (__ballot(cnt != 0) == 0) break;
            uint32_t quota = crowded ? (cnt < kTiledList / 64u ? cnt : kTiledList / 64u) : cnt;
            const uint32_t incl = waveInclusiveScan(quota);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t idx = incl - quota;
#pragma unroll
            for (int t = 0; t < TILES; t++) {
                while (hits[t] != 0 && quota != 0 && idx < kTiledList) {
                    list[idx] = (uint16_t)(((uint32_t)t << 10) | ((uint32_t)lane << 4) | (uint32_t)__builtin_ctz(hits[t]));    /* = offset of the position in the group */
                    idx++;
                    quota--;
                    hits[t] &= hits[t] - 1;
                }
            }
            const uint32_t listedNow = total < kTiledList ? total : kTiledList;
#if PFAC_TILED_STATS
            tsPasses++;
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t base = 0; base < listedNow; base += 64u * (uint32_t)WALKS) {
                uint32_t o[WALKS];
                int match[WALKS];
                bool alive[WALKS];
#pragma unroll
                for (int k = 0; k < WALKS; k++) {
                    const uint32_t e = base + (uint32_t)k * 64u + (uint32_t)lane;
                    alive[k] = e < listedNow;
                    o[k] = alive[k] ? (uint32_t)list[e] : 0u;
                }
                runWalks(o, alive, match);
                if (reduce) appendPairs(o, match);
                else {
                    bool found = false;
#pragma unroll
                    for (int k = 0; k < WALKS; k++) found |= match[k] != 0;
                    if (__ballot(found) != 0) {                /* one position in two thousand matches on the Snort-style stream: most sets store nothing */
                        if (kLdsResults) {
#pragma unroll
                            for (int k = 0; k < WALKS; k++)
                                if (match[k] != 0) res[o[k]] = match[k];
                        } else {
                            /* every load of these walks has been consumed; the wait is for the zero stores of a group none of whose
                             * walks left LDS (vmcnt counts vector memory in issue order on gfx9): the zero is in L2 before its patch */
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                            for (int k = 0; k < WALKS; k++)
                                if (match[k] != 0) outGroup[o[k]] = match[k];
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        if (kLdsResults && !reduce) {                          /* the tile's results, whole lines */
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (whole) {
#pragma unroll
                for (int k = 0; k < 4 * TILES; k++)
                    if ((uint32_t)k * 256u < hi) __builtin_nontemporal_store(reinterpret_cast<const i32x4 *>(res)[k * 64 + lane], reinterpret_cast<i32x4 *>(outGroup) + k * 64 + lane);
            } else {
                for (uint32_t p = lo + (uint32_t)lane; p < hi; p += 64u) outGroup[p] = res[p];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    };

    const uint64_t gid = (uint64_t)blockIdx.x * waves + wave, stride = (uint64_t)gridDim.x * waves;
    if (a.owned != 0) {
        /* positions [0, owned) of a.in; groups are cut from the 16-byte aligned address at or below a.in */
        const uint64_t head = reinterpret_cast<uintptr_t>(a.in) & 15u;
        const unsigned char *base16 = a.in - head;
        const uint64_t spanAll = (head + a.n + 15u) & ~uint64_t(15), ownEnd = head + a.owned, readEnd = head + a.n;
        const uint64_t groups = (ownEnd + kGroup - 1) / kGroup;
        for (uint64_t g = gid; g < groups; g += stride) {
            const uint64_t T = g * kGroup;
            const uint32_t lo = T < head ? (uint32_t)(head - T) : 0u;
            const uint32_t hi = ownEnd - T < kGroup ? (uint32_t)(ownEnd - T) : kGroup;
            const uint32_t limit = readEnd - T < kTiledFar ? (uint32_t)(readEnd - T) : kTiledFar;
            int *outGroup = reinterpret_cast<int *>(reinterpret_cast<uintptr_t>(a.out) + (T - head) * 4u);     /* T < head only in group 0, whose first `head` slots are never written */
            scanGroup(base16 + T, spanAll - T, lo, hi, limit, outGroup, a.reduceBase + (uint32_t)(T - head));
        }
    }
    if (listed != 0) {
        /* the chunks the filter kernel in front of this launch left to this kernel (ScanArgs::denseList): denseIn is the
         * 16-byte aligned first byte of that launch's input.  A chunk is smaller than a group: the rest of the group is masked */
        constexpr uint32_t kChunk = (uint32_t)kChunkBytesDev;
        constexpr uint32_t kPerChunk = kChunk > kGroup ? kChunk / kGroup : 1u, kTake = kChunk > kGroup ? kGroup : kChunk;
        const uint64_t spanAll = (a.denseReadable + 15u) & ~uint64_t(15);
        const uint64_t items = (uint64_t)listed * kPerChunk;
        for (uint64_t it = gid; it < items; it += stride) {
            const uint64_t T = (uint64_t)a.denseList[it / kPerChunk] * kChunk + (it % kPerChunk) * kGroup;
            const uint32_t limit = a.denseReadable - T < kTiledFar ? (uint32_t)(a.denseReadable - T) : kTiledFar;
            scanGroup(a.denseIn + T, spanAll - T, 0u, kTake, limit, a.denseOut + T, 0u);
        }
    }
    if (reduce) flushPairs();
    /* a whole big call through this kernel (PFACX_KERNEL_AUTO sent it here because the handle's last launch found its stream
     * pattern-dense): is it still?  Every wave adds its groups; the last one out tells the host and leaves the words zero */
#ifndef PFAC_NO_DENSE_REPORT
    if (a.reportDense != 0 && a.hostHint != nullptr) {
        /* through LDS first (the pair staging of the block's first wave, unused in a full-result launch): 4096 waves adding to one line of
         * device memory were 130 us of a 480 us launch; one wave per block does it for its block */
        if (lane == 0) {
            atomicAdd(&blockAcc[0], tsDense);
            atomicAdd(&blockAcc[1], tsGroups);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (atomicAdd(&blockAcc[2], 1u) == waves - 1u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                atomicAdd(a.work + pfac::kTiledDenseWord, blockAcc[0]);
                atomicAdd(a.work + pfac::kTiledDenseWord + 1, blockAcc[1]);
                __threadfence();
                if (atomicAdd(a.work + pfac::kTiledDenseWord + 2, 1u) == gridDim.x - 1u) {
                    __threadfence();
                    const unsigned int dense = atomicExch(a.work + pfac::kTiledDenseWord, 0u), all = atomicExch(a.work + pfac::kTiledDenseWord + 1, 0u);
                    atomicExch(a.work + pfac::kTiledDenseWord + 2, 0u);
                    __hip_atomic_store(a.hostHint + 1, dense * 2u > all ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
#endif
#if PFAC_TILED_STATS
    if (lane == 0) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(a.work + pfac::kStatsWord) + 26;   /* behind the PFAC_TIMING words */
        atomicAdd(acc + 0, (unsigned long long)tsIter); atomicAdd(acc + 1, (unsigned long long)tsLane); atomicAdd(acc + 2, (unsigned long long)tsWalks);
        atomicAdd(acc + 3, (unsigned long long)tsPasses); atomicAdd(acc + 4, (unsigned long long)tsDense); atomicAdd(acc + 5, (unsigned long long)tsGroups);
    }
#else
    (void)tsIter; (void)tsLane; (void)tsWalks; (void)tsPasses; (void)tsDense; (void)tsGroups;
#endif
}

/* ------------------------------------------------------------- launching */

/* the CU's 160 KiB: the prefilter bitmaps (<= kFilterLdsBudget, pattern_compiler.cpp) + control block + per scanning wave a
 * walk queue (24 B per entry), the staged chunk and the hit list (+ the pair staging of the compacted-output variant) */
constexpr size_t kLdsPerCu = 160 * 1024;
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

constexpr size_t kChunkBytesHost = (size_t)kGroupTiles * kTileBytes;
size_t chunkBytes(const PFAC_context *) { return kChunkBytesHost; }

/* hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the occupancy answer are per DEVICE state of one kernel
 * instantiation: a process that drives several GPUs (PFACX_matchFromHostMultiGPU: one thread and one handle per device)
 * must set the attribute on each of them.  It is set to the whole CU once per (instantiation, device), so that no launch
 * ever depends on what another handle with another pattern set asked for in between; the occupancy query runs with
 * that size (a 1024-thread block with 128 registers per thread fills a CU by itself whatever its LDS). */
constexpr int kMaxDevices = 64;
struct ShapeCache { std::mutex lock; int perCU[kMaxDevices] = {}; };

template <bool TEX, bool HAS_SHORT, bool REDUCE, bool STAGE>
hipError_t launchFilter(const PFAC_context *c, const ScanArgs &a0)
{
    auto kernel = pfac_scan_filter<TEX, HAS_SHORT, REDUCE, REDUCE ? PFAC_WALK_SETS : PFAC_WALK_SETS_FULL, STAGE>;
    static ShapeCache cache;
    size_t lds = filterLdsBytes(c, REDUCE, STAGE);
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

/* pfac_scan_tiled.  A launch that has whole megabytes in front of it -- or the dense-chunk list of a filter launch --
 * runs one persistent 1024-thread block per CU, groups of kTiledTilesBig KiB per wave, and every LDS byte the waves'
 * buffers leave as hot table rows; a small call runs 256-thread blocks, 1 KiB per wave, with the initial state's row
 * only (filling LDS is what a call of a few KiB pays for). */
#ifndef PFAC_TILED_BIG_MIB
#define PFAC_TILED_BIG_MIB 8                    /* 2 MiB: 10.8 us through the small shape, 21.6 through the big one; 4 MiB 18.5 / 22.2; 8 MiB 28.4 / 23.1 */
#endif
constexpr size_t kTiledBigBytes = size_t(PFAC_TILED_BIG_MIB) << 20;
template <bool TEX>
hipError_t launchTiled(const PFAC_context *c, ScanArgs a)
{
    auto kernelBig = pfac_scan_tiled<TEX, kTiledWalks, kTiledTilesBig, false>;
    auto kernelBigHot = pfac_scan_tiled<TEX, kTiledWalks, kTiledTilesBig, true>;     /* every bucket of the table fits the CU's LDS: no global path in the step */
#ifndef PFAC_TILED_WALKS_SMALL
#define PFAC_TILED_WALKS_SMALL 2
#endif
    auto kernelSmall = pfac_scan_tiled<TEX, PFAC_TILED_WALKS_SMALL, 1, false>;
    static ShapeCache cache;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidValue;
    {
        std::lock_guard<std::mutex> g(cache.lock);
        if (cache.perCU[dev] == 0) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernelBig), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernelBigHot), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernelSmall), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
            if (e != hipSuccess) return e;
            cache.perCU[dev] = 1;
        }
    }
    const size_t head = reinterpret_cast<uintptr_t>(a.in) & 15u;
    const bool big = a.denseList != nullptr || a.owned >= kTiledBigBytes;
    const size_t group = (big ? (size_t)kTiledTilesBig : 1) * kTiledTile;
    const size_t groups = a.owned ? (head + a.owned + group - 1) / group : 0;
    const unsigned threads = big ? 1024u : 256u;
    const size_t waves = threads / 64;
    const size_t cus = (size_t)(c->multiProcessorCount > 0 ? c->multiProcessorCount : 256);
    const size_t fixed = (size_t(1) << c->filter.log2Bits) / 8 + (size_t)pfac::kCharSet * sizeof(pfac::ChainSlot) + waves * tiledWaveLds(big ? kTiledTilesBig : 1);
    if (fixed > kLdsPerCu) return hipErrorInvalidValue;
    size_t hot = 0;
    if (big) {
        hot = (kLdsPerCu - fixed) / sizeof(pfac::ChainSlot);
        if (hot > a.rootRow) hot = a.rootRow;            /* the buckets lie in front of the initial state's row */
    }
    a.hotSlots = (uint32_t)hot;
    size_t blocks = (groups + waves - 1) / waves;
    if (a.denseList != nullptr || blocks > (big ? cus : cus * 16)) blocks = big ? cus : cus * 16;
    if (blocks < 1) blocks = 1;
    const size_t lds = fixed + hot * sizeof(pfac::ChainSlot);
#if PFAC_TILED_STATS
    (void)hipMemsetAsync(c->d_workCounters + pfac::kStatsWord + 52, 0, 6 * sizeof(unsigned long long), 0);
#endif
    if (big && hot == a.rootRow) hipLaunchKernelGGL(kernelBigHot, dim3((unsigned)blocks), dim3(threads), lds, 0, a);
    else if (big) hipLaunchKernelGGL(kernelBig, dim3((unsigned)blocks), dim3(threads), lds, 0, a);
    else hipLaunchKernelGGL(kernelSmall, dim3((unsigned)blocks), dim3(threads), lds, 0, a);
#if PFAC_TILED_STATS
    {
        unsigned long long t[6];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(t, c->d_workCounters + pfac::kStatsWord + 52, sizeof(t), hipMemcpyDeviceToHost);
        fprintf(stderr, "PFAC_TILED_STATS owned %zu: groups %llu (dense %llu) passes %llu walks %llu wave-steps %llu live lane-steps %llu: %.2f steps per walk, %.1f live lanes per wave-step of %d\n",
                a.owned, t[5], t[4], t[3], t[2], t[0], t[1], t[2] ? (double)t[1] / t[2] : 0.0, t[0] ? (double)t[1] / t[0] : 0.0, 64 * kTiledWalks);
    }
#endif
    return hipGetLastError();
}

/* PFACX_KERNEL_REFTABLE: the tiled frame over the reference-layout table of the perf mode (pfac_scan_tiled<..., REF = MODE>) */
template <int MODE>
hipError_t launchTiledRef(const PFAC_context *c, ScanArgs a)
{
    auto kernelBig = pfac_scan_tiled<false, kTiledWalks, kTiledTilesBig, false, MODE>;
    auto kernelSmall = pfac_scan_tiled<false, PFAC_TILED_WALKS_SMALL, 1, false, MODE>;
    static ShapeCache cache;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidValue;
    {
        std::lock_guard<std::mutex> g(cache.lock);
        if (cache.perCU[dev] == 0) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernelBig), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernelSmall), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
            if (e != hipSuccess) return e;
            cache.perCU[dev] = 1;
        }
    }
    const size_t head = reinterpret_cast<uintptr_t>(a.in) & 15u;
    const bool big = a.owned >= kTiledBigBytes;
    const size_t group = (big ? (size_t)kTiledTilesBig : 1) * kTiledTile;
    const size_t groups = a.owned ? (head + a.owned + group - 1) / group : 0;
    const unsigned threads = big ? 1024u : 256u;
    const size_t waves = threads / 64;
    const size_t cus = (size_t)(c->multiProcessorCount > 0 ? c->multiProcessorCount : 256);
    const size_t lds = (size_t(1) << c->filter.log2Bits) / 8 + (size_t)pfac::kCharSet * sizeof(pfac::ChainSlot) + waves * tiledWaveLds(big ? kTiledTilesBig : 1);
    if (lds > kLdsPerCu) return hipErrorInvalidValue;
    a.hotSlots = 0;
    a.denseList = nullptr;
    size_t blocks = (groups + waves - 1) / waves;
    if (blocks > (big ? cus : cus * 16)) blocks = big ? cus : cus * 16;
    if (blocks < 1) blocks = 1;
    if (big) hipLaunchKernelGGL(kernelBig, dim3((unsigned)blocks), dim3(threads), lds, 0, a);
    else hipLaunchKernelGGL(kernelSmall, dim3((unsigned)blocks), dim3(threads), lds, 0, a);
    return hipGetLastError();
}

template <int MODE>
hipError_t launchNaive(const PFAC_context *c, const ScanArgs &a)
{
    size_t blocks = (a.owned + 255) / 256;
    const size_t cap = (size_t)(c->multiProcessorCount > 0 ? c->multiProcessorCount : 256) * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(pfac_scan_naive<MODE>, dim3((unsigned)blocks), dim3(256), 0, 0, a);
    return hipGetLastError();
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
    if (!REDUCE) {
        stage = c->walker == PFACX_WALKER_STAGE ||
                (c->walker == PFACX_WALKER_AUTO && c->h_modeHint != nullptr && *static_cast<volatile const unsigned int *>(c->h_modeHint) != 0);
    }
#ifdef PFAC_QUICK      /* development builds (register / ISA inspection): the bench instances only */
    if (REDUCE || !tex) return hipErrorNotSupported;
    if (stage) return c->filter.hasShort ? launchFilter<true, true, false, true>(c, a) : launchFilter<true, false, false, true>(c, a);
    return c->filter.hasShort ? launchFilter<true, true, false, false>(c, a) : launchFilter<true, false, false, false>(c, a);
#else
    if (!REDUCE && stage) {
        if (tex) return c->filter.hasShort ? launchFilter<true, true, false, true>(c, a) : launchFilter<true, false, false, true>(c, a);
        return c->filter.hasShort ? launchFilter<false, true, false, true>(c, a) : launchFilter<false, false, false, true>(c, a);
    }
    if (tex) return c->filter.hasShort ? launchFilter<true, true, REDUCE, false>(c, a) : launchFilter<true, false, REDUCE, false>(c, a);
    return c->filter.hasShort ? launchFilter<false, true, REDUCE, false>(c, a) : launchFilter<false, false, REDUCE, false>(c, a);
#endif
}

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
    a.shortBits = c->d_shortBits;
    a.ladder = c->d_ladder;
    a.final3 = c->d_final3;
    a.log2Bits = c->filter.log2Bits;
    a.log2BitsLad = c->filter.log2BitsLad;
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
size_t filterLength(const PFAC_context *c, size_t first, size_t ownEnd, size_t inputSize, bool vectorOk)
{
    if (!vectorOk || c->kernelVariant == PFACX_KERNEL_NAIVE || c->kernelVariant == PFACX_KERNEL_REFTABLE) return 0;
    if (c->kernelVariant == PFACX_KERNEL_AUTO) {
        if (ownEnd - first < kSmallInput) return 0;        /* filling ~90 KiB of LDS tables per block costs more than scanning this */
        /* the handle's last big launch found most of its stream pattern-dense (short patterns over text, runs of a pattern byte):
         * the filter kernel would list nearly every chunk for the tiled kernel after testing it; the tiled kernel takes the call
         * alone (snort-length set with 1-byte patterns: 133 -> 166 GB/s) and reports when the stream stops being dense */
        if (c->h_modeHint != nullptr && static_cast<volatile const unsigned int *>(c->h_modeHint)[1] != 0) return 0;
    }
    const size_t margin = (size_t)c->fa.maxPatternLen + 64 + kWalkHalo;   /* a window load reads up to 35 bytes beyond a walk's deepest byte; the prefetch of a chunk reads the 64 (full-result kernel: kWalkHalo) bytes behind it */
    const size_t safeEnd = inputSize > margin ? inputSize - margin : 0;
    const size_t end = ownEnd < safeEnd ? ownEnd : safeEnd;
    return end > first ? (end - first) / chunkBytes(c) * chunkBytes(c) : 0;
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

hipError_t launchNaiveFor(const PFAC_context *c, bool hashed, bool tex, const ScanArgs &part)
{
#ifdef PFAC_REFTABLE_PER_BYTE          /* measurement builds: round 4's one-thread-per-byte kernel behind PFACX_KERNEL_REFTABLE */
    if (hashed) return tex ? launchNaive<HASH_BUFFER>(c, part) : launchNaive<HASH_GLOBAL>(c, part);
    return tex ? launchNaive<DENSE_BUFFER>(c, part) : launchNaive<DENSE_GLOBAL>(c, part);
#else
    if (hashed) return tex ? launchTiledRef<HASH_BUFFER>(c, part) : launchTiledRef<HASH_GLOBAL>(c, part);
    return tex ? launchTiledRef<DENSE_BUFFER>(c, part) : launchTiledRef<DENSE_GLOBAL>(c, part);
#endif
}

/* what is not the filter kernel's: the tiled kernel (chained table, both perf modes), or -- PFACX_KERNEL_REFTABLE -- the
 * reference-shaped kernel on the reference-layout table of the perf mode */
hipError_t launchSimple(const PFAC_context *c, bool hashed, bool tex, const ScanArgs &part)
{
    if (c->kernelVariant == PFACX_KERNEL_REFTABLE) return launchNaiveFor(c, hashed, tex, part);
    return tex ? launchTiled<true>(c, part) : launchTiled<false>(c, part);
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
            e = launchChained<false>(c, part, tex);
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
                e = tex ? launchTiled<true>(c, rest) : launchTiled<false>(c, rest);
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
            e = launchSimple(c, hashed, tex, rest);
        }
    }
    if (e == hipSuccess && !headDone) {                 /* the whole input is in front of the first aligned byte */
        ScanArgs part = a;
        part.owned = head;
        part.n = input_size;
        e = launchSimple(c, hashed, tex, part);
    }
    return e == hipSuccess ? PFAC_STATUS_SUCCESS : PFAC_STATUS_INTERNAL_ERROR;
}

/* ------------------------------------------------- compacted output: the pairs in position order */

/* The scan leaves (position, id) pairs in the order its walks finished.  Positions are DISTINCT keys below n, so they
 * are ordered with a counting pass over position bins and a rank inside each bin -- four short launches over the pairs
 * (~0.6 M for the bench stream, ~35 us together) instead of a general radix sort (rocPRIM Onesweep: a histogram and four
 * digit passes with decoupled look-back, 0.12 ms for the same pairs).  The number of pairs is read from the device
 * counter: the launches are queued behind the scan without the host knowing it.
 *   pfac_order_count    pairs per bin (bin = position >> shift, at most 2^16 bins).  64 consecutive pairs of the list
 *                       come from four flushes of scanning waves, i.e. from a handful of bins: one atomic per distinct
 *                       bin and wave, not per pair (the part sustains ~25 atomics per ns)
 *   pfac_order_offsets  exclusive prefix sum of the counters: block k sums everything in front of its 1024 counters
 *                       itself (at most 252 KiB, coalesced, from L2) -- no pass between blocks; lists the bins with
 *                       more than 64 pairs
 *   pfac_order_scatter  pair -> its bin's range of the scratch arrays (any order inside the bin); the lanes of a wave
 *                       that share a bin share one atomic, all of a wave's atomics are one instruction
 *   pfac_order_rank     one thread per pair: final place = bin start + number of smaller positions in the bin; then
 *                       the bins with more than 64 pairs, one block per listed bin: the bin's positions as a bitmap
 *                       in LDS, rank = set bits below
 * More pairs than the scratch arrays hold: every launch leaves at once, the host grows the scratch and queues them again.
 */
constexpr unsigned kOrderMaxBinsLog2 = 16;
constexpr unsigned kOrderMinShift = 6;           /* a bin of 64 positions holds at most 64 pairs */
constexpr unsigned kOrderMaxShift = 15;          /* int input_size < 2^31 */
constexpr unsigned kOrderBlockBins = 1024;       /* counters per block of pfac_order_offsets */
constexpr unsigned kOrderCrowded = 64;

struct OrderArgs {
    const unsigned int *posIn;     /* the scan's pairs, any order */
    const int *idIn;
    unsigned int *count;           /* device counter of the pairs (behind crowdedCount: one memset clears all counters of a call) */
    unsigned int capacity;         /* pairs posTmp / idTmp hold */
    unsigned int shift;
    unsigned int bins;
    unsigned int *counts;          /* bins rounded up to whole blocks (zeros), then the crowded bins' counter and the pairs' counter */
    unsigned int *cursor;          /* same length: start -> (after the scatter) end of each bin */
    unsigned int *crowdedCount;
    unsigned int *crowded;         /* bins */
    unsigned int *posTmp;
    int *idTmp;
    unsigned int *posOut;
    int *idOut;
};

__global__ __launch_bounds__(256) void pfac_order_count(OrderArgs o)
{
    const unsigned int count = *o.count;
    if (count > o.capacity) return;
    const unsigned int lane = threadIdx.x & 63u;
    for (unsigned int first = blockIdx.x * 256u + (threadIdx.x & ~63u); first < count; first += gridDim.x * 256u) {
        const bool has = first + lane < count;
        const unsigned int b = has ? o.posIn[first + lane] >> o.shift : 0xFFFFFFFFu;
        unsigned long long todo = __ballot(has);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned int b0 = (unsigned int)__builtin_amdgcn_readlane((int)b, leader);
            const unsigned long long same = __ballot(b == b0);
            if ((int)lane == leader) atomicAdd(&o.counts[b0], (unsigned int)__popcll(same));
            todo &= ~same;
        }
    }
}

/* sum over the block's 256 threads (every thread gets it) and the exclusive prefix of `own` among them */
__device__ __forceinline__ unsigned int blockScan256(unsigned int own, unsigned int *waveSum, unsigned int &total)
{
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned int incl = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int up = __shfl_up(incl, d);
        if ((int)lane >= d) incl += up;
    }
    __syncthreads();                                    /* waveSum may still be read from the previous call */
    if (lane == 63) waveSum[wave] = incl;
    __syncthreads();
    unsigned int before = 0;
    total = 0;
    for (unsigned int w = 0; w < 4; w++) {
        if (w < wave) before += waveSum[w];
        total += waveSum[w];
    }
    return before + incl - own;
}

__global__ __launch_bounds__(256) void pfac_order_offsets(OrderArgs o)
{
    __shared__ unsigned int waveSum[4];
    if (*o.count > o.capacity) return;
    const unsigned int t = threadIdx.x;
    const unsigned int firstQuad = blockIdx.x * (kOrderBlockBins / 4);
    const u32x4 *all = reinterpret_cast<const u32x4 *>(o.counts);
    unsigned int front = 0;
    unsigned int q = t;
    for (; q + 7u * 256u < firstQuad; q += 8u * 256u) {                 /* eight loads in flight: the last block reads 252 KiB */
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = all[q + k * 256u];
#pragma unroll
        for (int k = 0; k < 8; k++) front += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    for (; q < firstQuad; q += 256u) {
        const u32x4 v = all[q];
        front += v.x + v.y + v.z + v.w;
    }
    const u32x4 c = all[firstQuad + t];
    unsigned int base = 0, ignored = 0;
    (void)blockScan256(front, waveSum, base);
    unsigned int run = base + blockScan256(c.x + c.y + c.z + c.w, waveSum, ignored);
    const unsigned int bin = (firstQuad + t) * 4;
    if (c.x > kOrderCrowded) o.crowded[atomicAdd(o.crowdedCount, 1u)] = bin;
    if (c.y > kOrderCrowded) o.crowded[atomicAdd(o.crowdedCount, 1u)] = bin + 1;
    if (c.z > kOrderCrowded) o.crowded[atomicAdd(o.crowdedCount, 1u)] = bin + 2;
    if (c.w > kOrderCrowded) o.crowded[atomicAdd(o.crowdedCount, 1u)] = bin + 3;
    u32x4 s;
    s.x = run; run += c.x;
    s.y = run; run += c.y;
    s.z = run; run += c.z;
    s.w = run;
    reinterpret_cast<u32x4 *>(o.cursor)[firstQuad + t] = s;
}

__global__ __launch_bounds__(256) void pfac_order_scatter(OrderArgs o)
{
    const unsigned int count = *o.count;
    if (count > o.capacity) return;
    const unsigned int lane = threadIdx.x & 63u;
    for (unsigned int first = blockIdx.x * 256u + (threadIdx.x & ~63u); first < count; first += gridDim.x * 256u) {
        const bool has = first + lane < count;
        const unsigned int p = has ? o.posIn[first + lane] : 0xFFFFFFFFu;
        const int id = has ? o.idIn[first + lane] : 0;
        const unsigned int b = has ? p >> o.shift : 0xFFFFFFFFu;
        /* the lanes that share a bin: the first of them takes the bin's cursor forward for all */
        unsigned int myLeader = lane, myRank = 0, groupSize = 0;
        unsigned long long todo = __ballot(has);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned int b0 = (unsigned int)__builtin_amdgcn_readlane((int)b, leader);
            const unsigned long long same = __ballot(b == b0);
            if (b == b0) {
                myLeader = (unsigned int)leader;
                myRank = (unsigned int)__popcll(same & ((1ull << lane) - 1ull));
                groupSize = (unsigned int)__popcll(same);
            }
            todo &= ~same;
        }
        unsigned int base = 0;
        if (has && lane == myLeader) base = atomicAdd(&o.cursor[b], groupSize);
        base = (unsigned int)__shfl((int)base, (int)myLeader);
        if (has) { o.posTmp[base + myRank] = p; o.idTmp[base + myRank] = id; }
    }
}

__global__ __launch_bounds__(256) void pfac_order_rank(OrderArgs o)
{
    __shared__ unsigned int bits[1u << (kOrderMaxShift - 5)];      /* the positions of one crowded bin */
    __shared__ unsigned int below[1u << (kOrderMaxShift - 5)];     /* set bits in front of each word */
    __shared__ unsigned int waveSum[4];
    const unsigned int count = *o.count;
    if (count > o.capacity) return;
    const unsigned int listed = *o.crowdedCount;
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        const unsigned int p = o.posTmp[i];
        const int id = o.idTmp[i];
        const unsigned int b = p >> o.shift;
        const unsigned int end = o.cursor[b], start = b ? o.cursor[b - 1] : 0u;
        if (end - start > kOrderCrowded) continue;                       /* placed below */
        unsigned int rank = 0;
        for (unsigned int j = start; j < end; j++) rank += o.posTmp[j] < p ? 1u : 0u;
        o.posOut[start + rank] = p;
        o.idOut[start + rank] = id;
    }
    /* bins with more than 64 pairs, one block per listed bin (the same trip count for every thread of the block) */
    const unsigned int t = threadIdx.x;
    const unsigned int words = 1u << (o.shift - 5), mask = (1u << o.shift) - 1u;
    for (unsigned int k = blockIdx.x; k < listed; k += gridDim.x) {
        const unsigned int b = o.crowded[k];
        const unsigned int end = o.cursor[b], start = b ? o.cursor[b - 1] : 0u, c = end - start;
        for (unsigned int w = t; w < words; w += 256u) bits[w] = 0;
        __syncthreads();
        for (unsigned int e = t; e < c; e += 256u) {
            const unsigned int p = o.posTmp[start + e] & mask;
            atomicOr(&bits[p >> 5], 1u << (p & 31u));
        }
        __syncthreads();
        /* exclusive prefix of the words' population counts: `per` consecutive words per thread */
        const unsigned int per = words > 256u ? words / 256u : 1u;
        unsigned int own = 0;
        for (unsigned int j = 0; j < per; j++) {
            const unsigned int w = t * per + j;
            if (w < words) own += (unsigned int)__popc(bits[w]);
        }
        unsigned int ignored = 0;
        unsigned int run = blockScan256(own, waveSum, ignored);
        for (unsigned int j = 0; j < per; j++) {
            const unsigned int w = t * per + j;
            if (w < words) { below[w] = run; run += (unsigned int)__popc(bits[w]); }
        }
        __syncthreads();
        for (unsigned int e = t; e < c; e += 256u) {
            const unsigned int pos = o.posTmp[start + e], p = pos & mask;
            const unsigned int rank = below[p >> 5] + (unsigned int)__popc(bits[p >> 5] & ((1u << (p & 31u)) - 1u));
            o.posOut[start + rank] = pos;
            o.idOut[start + rank] = o.idTmp[start + e];
        }
        __syncthreads();
    }
}

unsigned int gridFor(const PFAC_context *c, size_t items)
{
    const size_t cap = (size_t)(c->multiProcessorCount > 0 ? c->multiProcessorCount : 256) * 8;
    const size_t blocks = (items + 255) / 256;
    return (unsigned int)(blocks < 1 ? 1 : blocks > cap ? cap : blocks);
}

/* grow-only device scratch of the compacted-output path, owned by the handle (the caller holds its lock) */
PFAC_status_t reduceScratch(PFAC_context *mc, size_t need, char **base)
{
    if (mc->reduceScratchBytes < need) {
        if (mc->d_reduceScratch) (void)hipFree(mc->d_reduceScratch);
        mc->d_reduceScratch = nullptr;
        mc->reduceScratchBytes = 0;
        const size_t grow = need + need / 2;
        if (hipMalloc(&mc->d_reduceScratch, grow) != hipSuccess) { (void)hipGetLastError(); mc->d_reduceScratch = nullptr; return PFAC_STATUS_CUDA_ALLOC_FAILED; }
        mc->reduceScratchBytes = grow;
    }
    *base = static_cast<char *>(mc->d_reduceScratch);
    return PFAC_STATUS_SUCCESS;
}

/* The ordering of a compacted-output call over n input bytes: the handle's scratch cut into the arrays of OrderArgs
 * (room for at least `pairs` pairs; all the scratch there is), then clearCounters() in front of the scan and order()
 * behind it, both asynchronous on the null stream. */
struct PairOrder {
    OrderArgs o{};
    size_t counterBytes = 0;

    PFAC_status_t plan(PFAC_context *mc, size_t n, size_t pairs, int *d_ids, int *d_pos)
    {
        unsigned int log2n = 1;
        while (log2n < 32 && (n - 1) >> log2n) log2n++;
        o.shift = log2n > kOrderMaxBinsLog2 + kOrderMinShift ? log2n - kOrderMaxBinsLog2 : kOrderMinShift;
        if (o.shift > kOrderMaxShift) return PFAC_STATUS_INTERNAL_ERROR;
        o.bins = (unsigned int)(((n - 1) >> o.shift) + 1);
        const size_t padded = ((size_t)o.bins + kOrderBlockBins - 1) / kOrderBlockBins * kOrderBlockBins;
        counterBytes = (padded * sizeof(unsigned int) + 2 * sizeof(unsigned int) + 255) / 256 * 256;
        const size_t cursorBytes = padded * sizeof(unsigned int);
        const size_t listBytes = ((size_t)o.bins * sizeof(unsigned int) + 255) / 256 * 256;
        const size_t fixed = counterBytes + cursorBytes + listBytes;
        const size_t arrayBytes = (pairs * sizeof(int) + 255) / 256 * 256;
        char *base = nullptr;
        const PFAC_status_t st = reduceScratch(mc, fixed + 2 * arrayBytes, &base);
        if (st != PFAC_STATUS_SUCCESS) return st;
        const size_t perArray = (mc->reduceScratchBytes - fixed) / 2 / 256 * 256;
        o.capacity = perArray / sizeof(int) > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned int)(perArray / sizeof(int));
        o.counts = reinterpret_cast<unsigned int *>(base);
        o.crowdedCount = o.counts + padded;
        o.count = o.crowdedCount + 1;
        o.cursor = reinterpret_cast<unsigned int *>(base + counterBytes);
        o.crowded = reinterpret_cast<unsigned int *>(base + counterBytes + cursorBytes);
        o.posTmp = reinterpret_cast<unsigned int *>(base + fixed);
        o.idTmp = reinterpret_cast<int *>(base + fixed + perArray);
        o.posIn = o.posOut = reinterpret_cast<unsigned int *>(d_pos);
        o.idIn = o.idOut = d_ids;
        return PFAC_STATUS_SUCCESS;
    }
    hipError_t clearCounters() const { return hipMemsetAsync(o.counts, 0, counterBytes, 0); }
    hipError_t order(const PFAC_context *c) const
    {
        const unsigned int grid = gridFor(c, o.capacity);
        hipLaunchKernelGGL(pfac_order_count, dim3(grid), dim3(256), 0, 0, o);
        hipLaunchKernelGGL(pfac_order_offsets, dim3((o.bins + kOrderBlockBins - 1) / kOrderBlockBins), dim3(256), 0, 0, o);
        hipLaunchKernelGGL(pfac_order_scatter, dim3(grid), dim3(256), 0, 0, o);
        hipLaunchKernelGGL(pfac_order_rank, dim3(grid), dim3(256), 0, 0, o);
        return hipGetLastError();
    }
};

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
    PairOrder order;
    const size_t expected = n / 128 > 65536 ? n / 128 : 65536;       /* room for one match per 128 bytes before the first call has been seen */
    st = order.plan(handle, n, expected, d_match_result, d_pos);
    if (st != PFAC_STATUS_SUCCESS) return st;
    if (order.clearCounters() != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
    const size_t head = headPositions(a.in, n);
    const size_t mainLen = filterLength(c, head, n, n, true);
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
        if (launchChained<true>(c, part, tex) != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
    } else {
        /* a small input (or PFACX_KERNEL_NAIVE / REFTABLE): positions [0, n) through the tiled (reference-shaped) kernel, which appends its matches to the list */
        ScanArgs part = a;
        part.owned = n;
        part.reduceBase = 0;
        if (launchSimple(c, hashed, tex, part) != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
    }
    if (ordered && order.order(c) != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
    unsigned int count = 0;
    if (hipMemcpy(&count, order.o.count, sizeof(count), hipMemcpyDeviceToHost) != hipSuccess) return PFAC_STATUS_INTERNAL_ERROR;
    if (count > (unsigned int)input_size) return PFAC_STATUS_INTERNAL_ERROR;
    if (ordered && count > order.o.capacity) {             /* more pairs than the scratch held: the launches left at once */
        st = order.plan(handle, n, count, d_match_result, d_pos);
        if (st != PFAC_STATUS_SUCCESS) return st;
        if (order.clearCounters() != hipSuccess || hipMemcpyAsync(order.o.count, &count, sizeof(count), hipMemcpyHostToDevice, 0) != hipSuccess ||
            order.order(c) != hipSuccess || hipStreamSynchronize(0) != hipSuccess)      /* `count` is read by that copy */
            return PFAC_STATUS_INTERNAL_ERROR;
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

/*
 * scan_common.h -- what the units of the kernel module libpfac_gfx950.so share: the kernel arguments (ScanArgs), the
 * reference-layout lookups (Lookup<MODE>), the slot helpers and walkers of the chained table (ChainCtx, ChainLane, StageLane,
 * boundedWalk), launch helpers.  Everything is in an unnamed namespace (each unit its own copy: inline device code); the few
 * functions one unit calls in another are declared in namespace pfacmod at the end.
 *   scan_filter.hip   pfac_scan_filter (the product kernel, both walkers, compacted output) + its launcher
 *   scan_tiled.hip    pfac_scan_tiled (chained table / reference-layout tables), pfac_scan_naive + launchers
 *   scan_order.hip    the four ordering kernels of the compacted output (PairOrder)
 *   scan_module.hip   the four symbols of the plugin seam (include/pfac_module.h), launch plans, stream probe
 */
#ifndef PFAC_SCAN_COMMON_H_
#define PFAC_SCAN_COMMON_H_
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "the kernel module is written for gfx950 (CDNA4): wave64, gfx9 waitcnt semantics, 160 KiB LDS"
#endif
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include <cstdint>
#include <type_traits>
#include <vector>

#include "pfac_context.h"

/* the kernels' arguments have external linkage: the units of the module hand them to each other's launchers */
namespace pfacmod {
using pfac::Int2;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
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
    const uint32_t *tail;                              /* the tail table (pfac::Filter), 3 words per slot, or null */
    int log2Tail;
    const uint32_t *ladder;
    const uint32_t *final3;
    const uint32_t *shortBits;
    int log2Bits, log2BitsLad, log2BitsF3;
    int ladderLast;                                    /* deepest level of the prefix ladder (pfac::Filter::ladderLast): behind kLadderLast only the VETO kernels look */
    uint32_t ladderSalt;                               /* pfac::Filter::ladderSalt */
    int skipCount;                                     /* skip tags (pfac::Filter): depth-6 ladder hashes whose candidates are next asked at kLadderLast */
    uint32_t skipTags[pfac::kSkipTagsMax];
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
}

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
using pfacmod::u32x4;

enum TableMode { DENSE_GLOBAL = 0, DENSE_BUFFER = 1, HASH_GLOBAL = 2, HASH_BUFFER = 3 };

using pfacmod::ScanArgs;
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

template <bool TEX, uint32_t ENTRY, bool SPEC = false> struct ChainLane {
    using Ctx = ChainCtx<TEX>;
    static constexpr bool kDeep = ENTRY > kEntryBytes;          /* 36-byte entries: the window is nine dwords, re-fetched 32 bytes at a time */
    static constexpr bool kSpec = kDeep && (SPEC || PFAC_WIDE_SPEC != 0); /* wide buckets: header and extension unit are fetched together, the unit compared out of the window */
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

/* ---- shared by the units of the module (scan_filter.hip, scan_tiled.hip, scan_order.hip, scan_module.hip) ---- */
constexpr size_t kLdsPerCu = 160 * 1024;
constexpr size_t kChunkBytesHost = (size_t)pfac::kChunkTiles * 1024;      /* input bytes of a chunk of the filter kernel */
constexpr size_t kChunkBytesDev = kChunkBytesHost;
/* hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the occupancy answer are per DEVICE state of one kernel
 * instantiation: a process that drives several GPUs (PFACX_matchFromHostMultiGPU: one thread and one handle per device)
 * must set the attribute on each of them.  It is set to the whole CU once per (instantiation, device), so that no launch
 * ever depends on what another handle with another pattern set asked for in between; the occupancy query runs with
 * that size (a 1024-thread block with 128 registers per thread fills a CU by itself whatever its LDS). */
constexpr int kMaxDevices = 64;
struct ShapeCache { std::mutex lock; int perCU[kMaxDevices] = {}; };

inline unsigned int gridFor(const PFAC_context *c, size_t items)
{
    const size_t cap = (size_t)(c->multiProcessorCount > 0 ? c->multiProcessorCount : 256) * 8;
    const size_t blocks = (items + 255) / 256;
    return (unsigned int)(blocks < 1 ? 1 : blocks > cap ? cap : blocks);
}

} // namespace

/* what one unit of the module calls in another (external linkage) */
namespace pfacmod {
/* scan_filter.hip: the filter kernel over a.n bytes (whole chunks); reduce = compacted output */
hipError_t launchFilterKernel(const PFAC_context *c, const ScanArgs &a, bool tex, bool reduce);
/* scan_tiled.hip: the tiled kernel over the chained table; and "whatever is not the filter kernel's" (PFACX_KERNEL_REFTABLE: over the reference-layout table of the perf mode) */
hipError_t launchTiledKernel(const PFAC_context *c, const ScanArgs &a, bool tex);
hipError_t launchSimpleKernel(const PFAC_context *c, bool hashed, bool tex, const ScanArgs &part);
/* ... the tiled frame over part.dense = int[S][256] whatever the perf mode and the kernel variant (PFAC_context::d_denseFast) */
hipError_t launchDenseTableKernel(const PFAC_context *c, bool tex, const ScanArgs &part);
}

#endif /* PFAC_SCAN_COMMON_H_ */

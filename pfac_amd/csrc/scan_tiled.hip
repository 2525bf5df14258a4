/*
 * scan_tiled.hip -- pfac_scan_tiled (small calls, PFACX_KERNEL_NAIVE, pattern-dense chunks and streams; with REF >= 0 the
 * reference-layout tables: PFACX_KERNEL_REFTABLE), pfac_scan_naive (measurement builds), and their launchers.
 * (part of the kernel module libpfac_gfx950.so: see scan_common.h and scan_module.hip)
 */
#include "scan_common.h"

namespace {

/* ------------------------------------------- reference-shaped kernel (PFACX_KERNEL_REFTABLE) */

/* One thread per input byte, no prefilter: the reference's algorithm with only the initial-state row
 * staged in LDS.  Alignment-agnostic, 64-bit positions.  Produces results for positions [0, owned);
 * walks may read up to a.n (owned <= n). */
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
constexpr int kRefWalksBig = 4;                        /* ... walks per lane (8: 128 VGPRs, C3 hashed 0.317 against 0.324 -- the rounds are bound by the number of gathers by then) */
constexpr uint32_t kRefPend = 64 * kRefWalksBig + 256;                      /* reference-table kernel, big shape: positions of several groups that wait for a walk round (32-bit offsets from a.in) */
constexpr int kRefTilesBig = 2;                        /* ... whose groups are 2 KiB: the walk rounds are filled from the pending list, not from one group, and the list needs the LDS */
constexpr uint32_t tiledWaveLds(int tiles, bool refBig = false) { return (uint32_t)tiles * kTiledTile + kTiledHalo + kTiledList * 2 + kTiledPairs * 8 + (tiles == 1 ? kTiledTile * 4 : 0) + (refBig ? kRefPend * 4 : 0); }

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
    /* the reference-layout tables cost one (dense) or two dependent (hashed) gathers per BYTE: in the big shape a position walks them only
     * if its first four bytes are a pattern prefix (prefix4: the trie's depth-4 nodes, two probes), or a pattern of three bytes (final3) or
     * of one or two (shortBits) matches there -- the level-4 test of the compacted-output filter kernel (pfac_context.h: struct Filter;
     * all three are supersets: a miss proves the result is 0).  Snort-style stream: 53.2 M -> 21.3 M walks per GiB.
     * And the walks of a sparse group do not run with the group: a step through these tables is a dependent gather (or two) per byte and a
     * round lasts as long as its deepest walk, so a round of 80 walks costs what a round of 256 does.  The survivors' positions go on a
     * pending list (kRefPend offsets from a.in per wave) and a round runs when 256 are waiting, its input bytes read from global memory
     * (they have just been streamed: L2), one byte ahead of the table lookups.  C3 through the hashed pair: 0.227 -> see DESIGN 3.2b. */
    constexpr bool kRefLevel4 = kRef && TILES > 1;
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
    unsigned char *waveBase = reinterpret_cast<unsigned char *>(sHot + a.hotSlots) + wave * tiledWaveLds(TILES, kRefLevel4);
    uint32_t *stage = reinterpret_cast<uint32_t *>(waveBase);
    uint16_t *list = reinterpret_cast<uint16_t *>(waveBase + kStage);
    constexpr bool kLdsResults = TILES == 1;               /* a sparse group's results are assembled in LDS and stored once, as whole lines */
    uint32_t *pairPos = reinterpret_cast<uint32_t *>(waveBase + kStage + kTiledList * 2), *pairId = pairPos + kTiledPairs;
    int *res = reinterpret_cast<int *>(waveBase + kStage + kTiledList * 2 + kTiledPairs * 8);
    uint32_t *pend = reinterpret_cast<uint32_t *>(waveBase + kStage + kTiledList * 2 + kTiledPairs * 8);      /* kRefLevel4 (TILES > 1: no `res`) */
    (void)pend;
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
            if constexpr (kRefLevel4) {
                /* the hot-row space (launchTiledRef sizes it): prefix4 | final3 | the exact 2-byte bitmap, if the set has patterns that short */
                const uint32_t p4 = (1u << pfac::kPrefix4Log2) / 128u, f3 = (1u << a.log2BitsF3) / 128u, sb = a.shortBits != nullptr ? 65536u / 128u : 0u;
                for (uint32_t i = (uint32_t)tid; i < p4; i += blockDim.x) sHot[i] = reinterpret_cast<const u32x4 *>(a.prefix4)[i];
                if ((1u << a.log2BitsF3) >= 128u) { for (uint32_t i = (uint32_t)tid; i < f3; i += blockDim.x) sHot[p4 + i] = reinterpret_cast<const u32x4 *>(a.final3)[i]; }
                else for (uint32_t i = (uint32_t)tid; i < (1u << a.log2BitsF3) / 32u; i += blockDim.x) reinterpret_cast<uint32_t *>(sHot + p4)[i] = a.final3[i];
                for (uint32_t i = (uint32_t)tid; i < sb; i += blockDim.x) sHot[p4 + (f3 ? f3 : 1u) + i] = reinterpret_cast<const u32x4 *>(a.shortBits)[i];
            }
        } else {
            for (int i = tid; i < pfac::kCharSet; i += (int)blockDim.x) sRoot[i] = a.chainSlots[a.rootRow + (uint32_t)i];
            for (uint32_t i = (uint32_t)tid; i < a.hotSlots; i += blockDim.x) sHot[i] = a.chainSlots[i];
        }
    }
    __syncthreads();
    const int *sInit = reinterpret_cast<const int *>(sRoot);
    (void)sInit;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(a.chainSlots), 0, (int)a.chainBytes, 0x00020000);
    const uint32_t mask3 = ((1u << ((uint32_t)a.log2Bits - 5u)) - 1u) << 2;      /* pfac::gram3Word as a byte address: (product >> 16) & mask3 */
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
    uint32_t tsIter = 0, tsLane = 0, tsWalks = 0, tsPasses = 0, tsDense = 0, tsGroups = 0, tsExt = 0, tsExtLanes = 0;
    (void)tsExt; (void)tsExtLanes;

    /* One group: g16 = its 16-byte aligned first byte; `span` bytes from there may be loaded (a multiple of 16: up to the
     * end of the 16-byte block that holds the last input byte); positions [lo, hi) of the group get a result, written to
     * outGroup[offset]; `limit` = group offset of the first byte behind the input (a pattern cannot reach it);
     * posBase = position of the group's first byte in the caller's stream (compacted output) */
    uint32_t pendCount = 0;                               /* wave-uniform */
    const bool deferOk = kRefLevel4 && a.n < 0xFFFFFF00ull && a.denseList == nullptr;    /* the pending list holds 32-bit offsets */
    /* compacted output: the matches of a walk set join the wave's staged pairs */
    auto appendPairsFrom = [&](uint32_t posBase, const uint32_t (&o)[WALKS], const int (&match)[WALKS]) __attribute__((always_inline)) {
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


    /* kRefLevel4: WALKS walks per lane from the offsets p[] (from a.in) through the reference-layout table, one byte per step; the bytes
     * come from global memory, each loaded one step before it is used (beside the lookups of the step in front of it) */
    auto runWalksGlobal = [&](const uint32_t (&p)[WALKS], bool (&alive)[WALKS], int (&match)[WALKS]) __attribute__((always_inline)) {
        if constexpr (kRefLevel4) {
            const Lookup<kRef ? REF : 0> lookup(a);
            const int *sInit = reinterpret_cast<const int *>(sRoot);
            const uint32_t n32 = (uint32_t)a.n;                  /* deferOk: fits */
            /* the input bytes off .. off+7 (any alignment; bytes at or behind the end of the input read as 0: no walk uses them).  A byte a
             * step cost the near-miss stream (walks 24 to 60 bytes deep) more address-path time than the table's own gathers. */
            typedef uint32_t u32_a1 __attribute__((aligned(1)));
            auto load8 = [&](uint32_t off, uint32_t &lo, uint32_t &hi) __attribute__((always_inline)) {
                lo = hi = 0;
                if (off + 8u <= n32 && off + 8u > off) {
                    lo = *reinterpret_cast<const u32_a1 *>(a.in + off);
                    hi = *reinterpret_cast<const u32_a1 *>(a.in + off + 4u);
                } else {
                    for (uint32_t j = 0; j < 8u; j++)
                        if (off + j < n32 && off + j >= off) { if (j < 4u) lo |= (uint32_t)a.in[off + j] << (8u * j); else hi |= (uint32_t)a.in[off + j] << (8u * (j - 4u)); }
                }
            };
            /* all walks of a round take their i-th byte in the same iteration: the window of eight bytes moves for all of them at once, and is
             * loaded one iteration before its first byte is needed */
            uint32_t w0[WALKS], w1[WALKS], n0[WALKS], n1[WALKS];
            int state[WALKS];
#pragma unroll
            for (int k = 0; k < WALKS; k++) {
                match[k] = 0; state[k] = kTrap; w0[k] = w1[k] = n0[k] = n1[k] = 0;
                if (alive[k]) load8(p[k], w0[k], w1[k]);
                if (alive[k]) state[k] = sInit[w0[k] & 0xFFu];
                alive[k] = alive[k] & (state[k] != kTrap);
                match[k] = (alive[k] && state[k] <= a.numFinal) ? state[k] : 0;
            }
            for (uint32_t i = 1;; i++) {                         /* wave-uniform: byte p + i */
                bool any = false;
#pragma unroll
                for (int k = 0; k < WALKS; k++) { alive[k] = alive[k] & (p[k] + i < n32) & (p[k] + i > p[k]); any |= alive[k]; }
                if (__ballot(any) == 0) break;
#if PFAC_TILED_STATS
                tsIter++;
#pragma unroll
                for (int k = 0; k < WALKS; k++) tsLane += (uint32_t)__popcll(__ballot(alive[k]));
#endif
                const uint32_t in8 = i & 7u;
                if (in8 == 7u) {
#pragma unroll
                    for (int k = 0; k < WALKS; k++) if (alive[k]) load8(p[k] + i + 1u, n0[k], n1[k]);
                }
                int next[WALKS];
#pragma unroll
                for (int k = 0; k < WALKS; k++) {
                    next[k] = kTrap;
                    const uint32_t ch = ((in8 < 4u ? w0[k] : w1[k]) >> (8u * (in8 & 3u))) & 0xFFu;
                    if (alive[k]) next[k] = lookup(state[k], (int)ch);
                }
#pragma unroll
                for (int k = 0; k < WALKS; k++) {
                    alive[k] = alive[k] & (next[k] != kTrap);
                    match[k] = (alive[k] && next[k] <= a.numFinal) ? next[k] : match[k];
                    state[k] = next[k];
                    if (in8 == 7u) { w0[k] = n0[k]; w1[k] = n1[k]; }
                }
            }
        }
    };
    /* rounds of 64 x WALKS pending positions while that many are waiting (all: whatever is left) */
    auto drainPend = [&](bool all) __attribute__((always_inline)) {
        if constexpr (kRefLevel4) {
            while (pendCount >= 64u * (uint32_t)WALKS || (all && pendCount != 0)) {
                const uint32_t take = pendCount < 64u * (uint32_t)WALKS ? pendCount : 64u * (uint32_t)WALKS, first = pendCount - take;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint32_t p[WALKS];
                int match[WALKS];
                bool alive[WALKS];
#pragma unroll
                for (int k = 0; k < WALKS; k++) {
                    const uint32_t e = first + (uint32_t)k * 64u + (uint32_t)lane;
                    alive[k] = e < pendCount;
                    p[k] = alive[k] ? pend[e] : 0u;
                }
#if PFAC_TILED_STATS
#pragma unroll
                for (int k = 0; k < WALKS; k++) tsWalks += (uint32_t)__popcll(__ballot(alive[k]));
#endif
                runWalksGlobal(p, alive, match);
                if (reduce) appendPairsFrom(a.reduceBase, p, match);
                else {
                    bool found = false;
#pragma unroll
                    for (int k = 0; k < WALKS; k++) found |= match[k] != 0;
                    if (__ballot(found) != 0) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       /* the zeros of these positions' groups are in L2 before their patches */
#pragma unroll
                        for (int k = 0; k < WALKS; k++)
                            if (match[k] != 0) a.out[p[k]] = match[k];
                    }
                }
                pendCount = first;
            }
        }
    };

    /* rel32: offset of the group's byte 0 from a.in (mod 2^32; deferred walks of the reference-table kernel) */
    auto scanGroup = [&](const unsigned char *g16, uint64_t span, uint32_t lo, uint32_t hi, uint32_t limit, int *outGroup, uint32_t posBase, uint32_t rel32) {
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
                    word[q] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((product >> 16) & mask3);
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
        /* kRefLevel4: may a pattern match at a position whose first four bytes are x?  (prefix4, final3, shortBits: pfac_context.h) */
        auto level4 = [&](uint32_t x) __attribute__((always_inline)) -> bool {
            const uint32_t *sP4 = reinterpret_cast<const uint32_t *>(sHot);
            const uint32_t f3Words = (1u << a.log2BitsF3) / 32u;
            const uint32_t *sF3 = sP4 + (1u << pfac::kPrefix4Log2) / 32u, *sShort = sF3 + (f3Words < 4u ? 4u : f3Words);
            const uint32_t h = x * pfac::kLadMul0;
            const uint32_t i1 = h >> (32 - pfac::kPrefix4Log2), i2 = (uint32_t)(h * pfac::kLadMulS) >> (32 - pfac::kPrefix4Log2);
            const uint32_t shiftF3 = 32u - (uint32_t)a.log2BitsF3;
            uint32_t pass = testBit(sP4, i1) & testBit(sP4, i2);
            pass |= testBit(sF3, (uint32_t)__umul24(x, pfac::kFinal3Mul) >> shiftF3) & testBit(sF3, (uint32_t)__umul24(x, pfac::kFinal3Mul2) >> shiftF3);
            if (a.shortBits != nullptr) pass |= testBit(sShort, x & 0xFFFFu);
            return pass != 0;
        };
        (void)level4;
        auto runWalksRef = [&](const uint32_t (&o)[WALKS], bool (&alive)[WALKS], int (&match)[WALKS]) {
            if constexpr (kRef) {
                const Lookup<kRef ? REF : 0> lookup(a);
                uint32_t q[WALKS];
                int state[WALKS];
#pragma unroll
                for (int k = 0; k < WALKS; k++) {
                    q[k] = o[k]; match[k] = 0; state[k] = kTrap;
                    if constexpr (kRefLevel4) {
                        uint32_t x = 0, x1 = 0;
                        if (alive[k]) fetch(q[k], x, x1);
                        alive[k] = alive[k] & level4(x);
                    }
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
#if PFAC_TILED_STATS
                        tsExt++; tsExtLanes += (uint32_t)__popcll(__ballot(ok & isLong));
#endif
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
        auto appendPairs = [&](const uint32_t (&o)[WALKS], const int (&match)[WALKS]) { appendPairsFrom(posBase, o, match); };

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
            if constexpr (kRefLevel4) {
                if (deferOk) {
                    /* the level-4 test on the staged bytes; what passes waits on the pending list for a full round (at most 255 are
                     * waiting when a pass of at most kTiledList begins: kRefPend holds them) */
                    static_assert(64u * (uint32_t)WALKS + kTiledList <= kRefPend + 1u, "pending list");
                    for (uint32_t base = 0; base < listedNow; base += 64u) {
                        const uint32_t e = base + (uint32_t)lane;
                        const bool act = e < listedNow;
                        const uint32_t code = act ? (uint32_t)list[e] : 0u;
                        uint32_t x = 0, x1 = 0;
                        if (act) fetch(code, x, x1);
                        const bool keep = act && level4(x);
                        const uint64_t m = __ballot(keep);
                        if (keep) pend[pendCount + laneRankIn(m)] = rel32 + code;
                        pendCount += (uint32_t)__popcll(m);
                    }
                    drainPend(false);
                    continue;
                }
            }
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
            scanGroup(base16 + T, spanAll - T, lo, hi, limit, outGroup, a.reduceBase + (uint32_t)(T - head), (uint32_t)(T - head));
        }
        drainPend(true);
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
            scanGroup(a.denseIn + T, spanAll - T, 0u, kTake, limit, a.denseOut + T, 0u, 0u);
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
        atomicAdd(acc - 2, (unsigned long long)tsExt); atomicAdd(acc - 1, (unsigned long long)tsExtLanes);      /* the last two PFAC_TIMING words (the 8.25 KiB of counters end behind acc[5]) */
    }
#else
    (void)tsIter; (void)tsLane; (void)tsWalks; (void)tsPasses; (void)tsDense; (void)tsGroups;
#endif
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
#ifdef PFAC_EXP_HOT_ENV
        { static const char *cap = getenv("PFAC_EXP_HOT_KIB"); if (cap && a.denseList == nullptr) { const size_t k = (size_t)atoi(cap) * 1024 / sizeof(pfac::ChainSlot); if (hot > k) hot = k; } }
#endif
    }
    a.hotSlots = (uint32_t)hot;
    size_t blocks = (groups + waves - 1) / waves;
    if (a.denseList != nullptr || blocks > (big ? cus : cus * 16)) blocks = big ? cus : cus * 16;
    if (blocks < 1) blocks = 1;
    const size_t lds = fixed + hot * sizeof(pfac::ChainSlot);
#if PFAC_TILED_STATS
    (void)hipMemsetAsync(c->d_workCounters + pfac::kStatsWord + 48, 0, 8 * sizeof(unsigned long long), 0);
#endif
    if (big && hot == a.rootRow) hipLaunchKernelGGL(kernelBigHot, dim3((unsigned)blocks), dim3(threads), lds, 0, a);
    else if (big) hipLaunchKernelGGL(kernelBig, dim3((unsigned)blocks), dim3(threads), lds, 0, a);
    else hipLaunchKernelGGL(kernelSmall, dim3((unsigned)blocks), dim3(threads), lds, 0, a);
#if PFAC_TILED_STATS
    {
        unsigned long long t[8];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(t, c->d_workCounters + pfac::kStatsWord + 48, sizeof(t), hipMemcpyDeviceToHost);
        fprintf(stderr, "PFAC_TILED_STATS owned %zu: groups %llu (dense %llu) passes %llu walks %llu wave-steps %llu live lane-steps %llu: %.2f steps per walk, %.1f live lanes per wave-step of %d; long-slot unit fetches: %llu wave-level, %llu lanes\n",
                a.owned, t[7], t[6], t[5], t[4], t[2], t[3], t[4] ? (double)t[3] / t[4] : 0.0, t[2] ? (double)t[3] / t[2] : 0.0, 64 * kTiledWalks, t[0], t[1]);
    }
#endif
    return hipGetLastError();
}

/* PFACX_KERNEL_REFTABLE: the tiled frame over the reference-layout table of the perf mode (pfac_scan_tiled<..., REF = MODE>) */
template <int MODE>
hipError_t launchTiledRef(const PFAC_context *c, ScanArgs a)
{
    auto kernelBig = pfac_scan_tiled<false, kRefWalksBig, kRefTilesBig, false, MODE>;
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
    const size_t group = (big ? (size_t)kRefTilesBig : 1) * kTiledTile;
    const size_t groups = a.owned ? (head + a.owned + group - 1) / group : 0;
    const unsigned threads = big ? 1024u : 256u;
    const size_t waves = threads / 64;
    const size_t cus = (size_t)(c->multiProcessorCount > 0 ? c->multiProcessorCount : 256);
    /* big shape: prefix4 | final3 | shortBits in the place of the hot rows (pfac_scan_tiled: kRefLevel4), a pending list per wave */
    const size_t f3Bytes = (size_t(1) << c->filter.log2BitsF3) / 8;
    const size_t level4 = big ? (size_t(1) << pfac::kPrefix4Log2) / 8 + (f3Bytes < 16 ? 16 : f3Bytes) + (a.shortBits != nullptr ? 65536 / 8 : 0) : 0;
    const size_t lds = (size_t(1) << c->filter.log2Bits) / 8 + (size_t)pfac::kCharSet * sizeof(pfac::ChainSlot) + level4 + waves * tiledWaveLds(big ? kRefTilesBig : 1, big);
    if (lds > kLdsPerCu || (big && (a.prefix4 == nullptr || a.final3 == nullptr))) return hipErrorInvalidValue;
    a.hotSlots = (uint32_t)(level4 / 16);
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

} // namespace

namespace pfacmod {
hipError_t launchTiledKernel(const PFAC_context *c, const ScanArgs &a, bool tex) { return tex ? launchTiled<true>(c, a) : launchTiled<false>(c, a); }
hipError_t launchSimpleKernel(const PFAC_context *c, bool hashed, bool tex, const ScanArgs &part) { return launchSimple(c, hashed, tex, part); }
hipError_t launchDenseTableKernel(const PFAC_context *c, bool tex, const ScanArgs &part) { return tex ? launchTiledRef<DENSE_BUFFER>(c, part) : launchTiledRef<DENSE_GLOBAL>(c, part); }
}

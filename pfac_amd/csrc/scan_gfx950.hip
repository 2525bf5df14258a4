/*
 * scan_gfx950.hip -- kernel module libpfac_gfx950.so: the PFAC match path for
 * CDNA4 (MI355X).  Hand-written HIP for gfx950 only.
 *
 * Replaces the reference's PFAC_kernel_timeDriven / PFAC_kernel_spaceDriven
 * (PFAC/src/PFAC_kernel.cu:377-458, PFAC/src/PFAC_kernel_spaceDriven.cu:465-558)
 * and their host wrappers (:90-244 / :149-348).  Result contract is identical:
 * d_matched_result[j] = ID of the longest pattern starting at byte j, else 0,
 * every element written.
 *
 * Design (DESIGN.md has the numbers):
 *
 *   The path is HBM-bound integer work: 1 B read + 4 B written per input byte.
 *   The reference walks the automaton from every byte; on MI355X that makes the
 *   per-CU texture-address pipe (one gathered table line per lane per step),
 *   not HBM, the limit.  Here the walk is split:
 *
 *   1. FILTER  (all lanes, LDS only).  A wave owns a 1 KiB tile.  Lane l loads
 *      dword k*64+l of the tile for k=0..3 (four fully coalesced 256 B loads),
 *      gets the following dword from lane l+1, and tests each of its 16 start
 *      positions against a 3-gram Bloom bitmap held in LDS (plus an exact
 *      2-gram bitmap when patterns shorter than 3 bytes exist).  A position
 *      that misses cannot match anything, so its result is 0.
 *   2. ZERO STORES.  The tile's 4 KiB of results are written as 16 B/lane
 *      non-temporal stores, 1 KiB contiguous per wave instruction, with no
 *      dependence on the input.
 *   3. WALK  (compacted).  Surviving positions (a few %) are compacted into a
 *      per-wave LDS queue with ballot/mbcnt; the queue accumulates over up to
 *      64 tiles and is drained by 64 walker lanes, one position per lane:
 *      first transition from the initial-state row in LDS, the rest from the
 *      dense or hashed table in global memory (plain loads or buffer-resource
 *      loads = the "texture" mode).  A lane whose walk hits the trap state
 *      immediately takes the next queue entry (ballot + mbcnt hand out the
 *      entries), so lanes stay busy although walk depths differ; one wave
 *      ballot ends the drain when every lane is dead and the queue is empty.
 *      Before a level-1 survivor is queued its first four bytes are tested
 *      against a 4-gram bitmap in LDS (second filter level, only executed by
 *      the few lanes that hold a hit); the queue keeps those four bytes, so a
 *      walker needs no input load for its first four transitions.
 *      The hashed mode walks a device-side "fat" copy of the reference's hash
 *      table in which every slot also carries the row descriptor of its next
 *      state: one dependent 16 B load per transition instead of two.
 *      Non-zero results are stored after the wave has drained its zero stores
 *      (s_waitcnt vmcnt(0)), so they land on top.
 *
 *   Blocks are persistent (grid = CUs x resident blocks) and stride over tiles,
 *   so the LDS tables are filled once per block.  No MFMA: nothing here is a
 *   contraction.
 *
 * A second, deliberately simple kernel (one thread per byte, byte loads,
 * scalar stores) serves pointers the vector path cannot take (input not
 * 4-byte aligned, output not 16-byte aligned) and is the A/B baseline.
 */
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pfac_context.h"

namespace {

using pfac::Int2;

constexpr int kTrap = pfac::kTrapState;
constexpr int kBlockThreads = 1024;
constexpr int kWavesPerBlock = kBlockThreads / 64;
constexpr int kTileBytes = 1024;              /* input bytes per wave per iteration   */

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum TableMode { DENSE_GLOBAL = 0, DENSE_BUFFER = 1, HASH_GLOBAL = 2, HASH_BUFFER = 3 };

struct ScanArgs {
    const unsigned char *in;
    int *out;
    size_t n;
    const int *dense;
    const Int2 *hashRow;
    const Int2 *hashVal;
    const i32x4 *hashFat;                              /* {next, ch, next.offset, next.k|S-1} per slot */
    uint32_t denseBytes, hashRowBytes, hashValBytes, hashFatBytes;   /* buffer-resource extents */
    const int *initialRow;
    const Int2 *initialRowInfo;                        /* hashed: row descriptor of initialRow[c] */
    const uint32_t *gram3;
    const uint32_t *gram4;
    const uint32_t *final3;
    const uint32_t *shortBits;
    int log2Bits, log2Bits4, log2BitsF3;
    int numFinal;
    int initialState;
};

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

/* ------------------------------------------------------------ walk policies */

/* Cursor of one walker lane.  `off`/`ks` are only live in the hashed modes: they are the
 * reference's hashRowPtr entry {offset, (k<<16)|(S-1)} of the CURRENT state, carried along so a
 * transition needs one dependent load (the fat slot) instead of two (rowPtr, then valPtr). */
struct Cursor { int state, off, ks; };

template <int MODE> struct Walk;

template <> struct Walk<DENSE_GLOBAL> {
    const int *table; const int *sInit;
    __device__ Walk(const ScanArgs &a, const int *init, const Int2 *) : table(a.dense), sInit(init) {}
    __device__ __forceinline__ bool first(Cursor &c, int ch) const { c.state = sInit[ch]; return c.state != kTrap; }
    __device__ __forceinline__ bool step(Cursor &c, int ch) const
    {
        const int s = table[(size_t)(uint32_t)c.state * pfac::kCharSet + (uint32_t)ch];
        if (s == kTrap) return false;
        c.state = s;
        return true;
    }
};

template <> struct Walk<DENSE_BUFFER> {
    __amdgpu_buffer_rsrc_t rsrc; const int *sInit;
    __device__ Walk(const ScanArgs &a, const int *init, const Int2 *)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(a.dense), 0, (int)a.denseBytes, 0x00020000)), sInit(init) {}
    __device__ __forceinline__ bool first(Cursor &c, int ch) const { c.state = sInit[ch]; return c.state != kTrap; }
    __device__ __forceinline__ bool step(Cursor &c, int ch) const
    {
        const uint32_t off = ((uint32_t)c.state * pfac::kCharSet + (uint32_t)ch) * 4u;
        const int s = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
        if (s == kTrap || s == 0) return false;        /* 0 = out-of-range clamp: never a valid next state */
        c.state = s;
        return true;
    }
};

template <> struct Walk<HASH_GLOBAL> {
    const i32x4 *fat; const int *sInit; const Int2 *sInitRow;
    __device__ Walk(const ScanArgs &a, const int *init, const Int2 *initRow) : fat(a.hashFat), sInit(init), sInitRow(initRow) {}
    __device__ __forceinline__ bool first(Cursor &c, int ch) const
    {
        c.state = sInit[ch];
        const Int2 r = sInitRow[ch];
        c.off = r.x; c.ks = r.y;
        return c.state != kTrap;
    }
    __device__ __forceinline__ bool step(Cursor &c, int ch) const
    {
        if (c.off < 0) return false;                   /* state without outgoing transitions */
        const i32x4 v = fat[(uint32_t)c.off + (uint32_t)hashSlot(c.ks, ch)];
        if (v.y != ch) return false;                   /* empty slot (ch = -1) or another byte's slot */
        c.state = v.x; c.off = v.z; c.ks = v.w;
        return true;
    }
};

template <> struct Walk<HASH_BUFFER> {
    __amdgpu_buffer_rsrc_t rsrc; const int *sInit; const Int2 *sInitRow;
    __device__ Walk(const ScanArgs &a, const int *init, const Int2 *initRow)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<i32x4 *>(a.hashFat), 0, (int)a.hashFatBytes, 0x00020000)),
          sInit(init), sInitRow(initRow) {}
    __device__ __forceinline__ bool first(Cursor &c, int ch) const
    {
        c.state = sInit[ch];
        const Int2 r = sInitRow[ch];
        c.off = r.x; c.ks = r.y;
        return c.state != kTrap;
    }
    __device__ __forceinline__ bool step(Cursor &c, int ch) const
    {
        if (c.off < 0) return false;
        const uint32_t slot = (uint32_t)c.off + (uint32_t)hashSlot(c.ks, ch);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(slot * 16u), 0, 0);
        if ((int)v.y != ch) return false;
        c.state = (int)v.x; c.off = (int)v.z; c.ks = (int)v.w;
        return true;
    }
};

/* --------------------------------------------------------- filter kernel */

constexpr uint32_t kQueueCap = 512;           /* entries per wave                                     */
constexpr uint32_t kMaxSlots = 64;            /* tiles a queue may span: 6 bits of the position entry */

__device__ __forceinline__ uint32_t laneRankIn(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t testBit(const uint32_t *bitmap, uint32_t h) { return (bitmap[h >> 5] >> (h & 31)) & 1u; }

/* LDS view of one block */
struct Lds {
    const uint32_t *gram3, *gram4, *final3, *shortBits;
    const int *init;
    const Int2 *initRow;
    uint32_t shift3, shift4, shiftF3;
};

/* 8 input bytes starting at byte `pos` from aligned dword loads (input base is 4-byte aligned on
 * this path); dwords at or beyond numDwords read as 0. */
__device__ __forceinline__ uint64_t loadWindowAligned(const uint32_t *in32, size_t pos, size_t numDwords)
{
    const size_t w = pos >> 2;
    const uint32_t sh = (uint32_t)pos & 3u;
    const uint32_t w0 = w < numDwords ? in32[w] : 0u;
    const uint32_t w1 = w + 1 < numDwords ? in32[w + 1] : 0u;
    const uint32_t w2 = w + 2 < numDwords ? in32[w + 2] : 0u;
    const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
    const uint32_t hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
    return ((uint64_t)hi << 32) | lo;
}

/* Walk every queued position.  Entry i is byte (tile0 + (qPos[i]>>10)*tileStride)*1024 + (qPos[i]&1023)
 * and qBytes[i] holds its first four input bytes.  Lanes are refilled from the queue as soon as
 * their walk ends. */
template <int MODE, bool HAS_SHORT>
__device__ __forceinline__ void drainQueue(const ScanArgs &a, const Walk<MODE> &walk, const Lds &,
                                           const uint16_t *qPos, const uint32_t *qBytes, uint32_t qn,
                                           size_t tile0, size_t tileStride, size_t numDwords)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const size_t n = a.n;
    const uint32_t *in32 = reinterpret_cast<const uint32_t *>(a.in);
    uint32_t qhead = 0;                         /* wave-uniform */
    bool alive = false;
    size_t pos = 0;
    Cursor cur{kTrap, -1, -1};
    int match = 0;
    uint64_t win = 0;
    uint32_t depth = 0;
    for (;;) {
        const uint64_t idle = __ballot(!alive);
        if (idle && qhead < qn) {
            const uint32_t idx = qhead + laneRankIn(idle);
            if (!alive && idx < qn) {
                const uint32_t x = qBytes[idx];
                const uint32_t e = qPos[idx];
                pos = (tile0 + (size_t)(e >> 10) * tileStride) * kTileBytes + (e & 1023u);
                alive = walk.first(cur, (int)(x & 0xFF));     /* ref phi_s02s1, PFAC_kernel.cu:259 */
                match = (alive && cur.state <= a.numFinal) ? cur.state : 0;
                win = x >> 8;
                depth = 1;
            }
            qhead += (uint32_t)__popcll(idle);
        }
        if (!__ballot(alive)) {                 /* every lane dead: done when the queue is empty too */
            if (qhead >= qn) break;
            continue;
        }
        if (alive) {
            if ((depth & 7u) == 4u) win = loadWindowAligned(in32, pos + depth, numDwords);
            const int ch = (int)(win & 0xFF);
            win >>= 8;
            if (pos + depth < n && walk.step(cur, ch)) {
                if (cur.state <= a.numFinal) match = cur.state;
                depth++;
            } else {
                alive = false;
                if (match != 0) {
                    /* the zero stores covering this position were issued earlier by this wave;
                     * they must have reached L2 before the patch goes out */
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    a.out[pos] = match;
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   /* queue memory is reused */
}

template <int MODE, bool HAS_SHORT>
__global__ __launch_bounds__(kBlockThreads) void pfac_scan_filter(ScanArgs a)
{
    constexpr bool kHashed = (MODE == HASH_GLOBAL || MODE == HASH_BUFFER);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int words3 = 1 << (a.log2Bits - 5), words4 = 1 << (a.log2Bits4 - 5), wordsF3 = 1 << (a.log2BitsF3 - 5);
    uint32_t *sGram3 = reinterpret_cast<uint32_t *>(smem);
    uint32_t *sGram4 = sGram3 + words3;
    uint32_t *sFinal3 = sGram4 + words4;
    uint32_t *sShort = sFinal3 + wordsF3;
    int *sInit = reinterpret_cast<int *>(sShort + (HAS_SHORT ? 2048 : 0));
    Int2 *sInitRow = reinterpret_cast<Int2 *>(sInit + pfac::kCharSet);
    uint32_t *sQBytesAll = reinterpret_cast<uint32_t *>(sInitRow + (kHashed ? pfac::kCharSet : 0));
    uint16_t *sQPosAll = reinterpret_cast<uint16_t *>(sQBytesAll + kWavesPerBlock * kQueueCap);

    const int tid = threadIdx.x;
    {   /* fill the LDS tables once per (persistent) block, 16 B per lane */
        auto copy16 = [&](uint32_t *dst, const uint32_t *src, int words) {
            const u32x4 *g = reinterpret_cast<const u32x4 *>(src);
            u32x4 *s = reinterpret_cast<u32x4 *>(dst);
            for (int i = tid; i < words / 4; i += kBlockThreads) s[i] = g[i];
        };
        copy16(sGram3, a.gram3, words3);
        copy16(sGram4, a.gram4, words4);
        copy16(sFinal3, a.final3, wordsF3);
        if (HAS_SHORT) copy16(sShort, a.shortBits, 2048);
        if (tid < pfac::kCharSet) {
            sInit[tid] = a.initialRow[tid];
            if (kHashed) sInitRow[tid] = a.initialRowInfo[tid];
        }
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = tid >> 6;
    uint32_t *qBytes = sQBytesAll + wave * kQueueCap;
    uint16_t *qPos = sQPosAll + wave * kQueueCap;
    const Walk<MODE> walk(a, sInit, sInitRow);
    const Lds lds{sGram3, sGram4, sFinal3, sShort, sInit, sInitRow,
                  32u - (uint32_t)a.log2Bits, 32u - (uint32_t)a.log2Bits4, 32u - (uint32_t)a.log2BitsF3};
    const uint32_t *in32 = reinterpret_cast<const uint32_t *>(a.in);
    const size_t n = a.n;
    const size_t numTiles = (n + kTileBytes - 1) / kTileBytes;
    const size_t numDwords = (n + 3) >> 2;          /* dwords that may be read (reference pads the same way, PFAC.cpp:838-842) */
    const size_t totalWaves = (size_t)gridDim.x * kWavesPerBlock;

    uint32_t qn = 0;                                /* queued positions (wave-uniform)             */
    uint32_t slot = 0;                              /* tiles since the queue was last empty        */
    size_t tile0 = 0;                               /* tile of slot 0                              */

    for (size_t tile = (size_t)blockIdx.x * kWavesPerBlock + wave; tile < numTiles; tile += totalWaves) {
        const size_t base = tile * kTileBytes;
        const size_t dwBase = tile * (kTileBytes / 4);
        const bool full = base + kTileBytes <= n;            /* wave-uniform */
        if (slot == 0) tile0 = tile;

        /* ---- 1. input: 4 coalesced dword loads per lane + 1 halo dword per wave */
        uint32_t d[4];
        uint32_t halo = 0;
        if (full) {
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = in32[dwBase + k * 64 + lane];
            if (dwBase + 256 < numDwords) halo = in32[dwBase + 256];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const size_t idx = dwBase + k * 64 + lane;
                d[k] = idx < numDwords ? in32[idx] : 0u;
            }
        }

        /* ---- 2. zero stores: 16 B per lane, 1 KiB contiguous per instruction */
        if (full) {
            i32x4 *o4 = reinterpret_cast<i32x4 *>(a.out + base);
            const i32x4 zero = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; k++) __builtin_nontemporal_store(zero, &o4[k * 64 + lane]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const size_t p0 = base + k * 256 + lane * 4;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (p0 + i < n) a.out[p0 + i] = 0;
            }
        }

        /* ---- 3. filter level 1: one LDS bit test per start position */
        uint32_t hits = 0;
        uint32_t nxt[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            nxt[k] = (uint32_t)__shfl_down((int)d[k], 1);
            const uint32_t wrap = (k < 3) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)d[(k + 1) & 3]) : halo;
            if (lane == 63) nxt[k] = wrap;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t x = __builtin_amdgcn_alignbyte(nxt[k], d[k], i);   /* bytes pos..pos+3 */
                /* __umul24 returns int: the shift must be logical */
                uint32_t bit = testBit(sGram3, (uint32_t)__umul24(x, pfac::kGram3Mul) >> lds.shift3);
                if (HAS_SHORT) bit |= testBit(sShort, x & 0xFFFFu);
                hits |= bit << (k * 4 + i);
            }
        }
        if (!full) {   /* never queue a position at or beyond n */
            uint32_t valid = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (base + k * 256 + lane * 4 + i < n) valid |= 1u << (k * 4 + i);
            hits &= valid;
        }

        /* ---- 4. append surviving positions (+ their first 4 bytes) to the wave's queue */
        uint64_t pending = __ballot(hits != 0);
        while (pending) {                                   /* wave-uniform: max hits per lane iterations */
            if (qn + 64 > kQueueCap) {                      /* full: walk what is queued, restart at this tile */
                drainQueue<MODE, HAS_SHORT>(a, walk, lds, qPos, qBytes, qn, tile0, totalWaves, numDwords);
                qn = 0; slot = 0; tile0 = tile;
            }
            const bool has = hits != 0;
            const uint32_t b = (uint32_t)__builtin_ctz(hits | 0x10000u);
            const uint32_t k = b >> 2;
            const uint32_t dk = k == 0 ? d[0] : k == 1 ? d[1] : k == 2 ? d[2] : d[3];
            const uint32_t nk = k == 0 ? nxt[0] : k == 1 ? nxt[1] : k == 2 ? nxt[2] : nxt[3];
            const uint32_t x = __builtin_amdgcn_alignbyte(nk, dk, b & 3);
            /* filter level 2 (only lanes holding a level-1 hit): the walk survives four transitions,
             * or a pattern of length <= 3 can match here */
            bool keep = false;
            if (has) {
                uint32_t pass = testBit(sGram4, (x * pfac::kGram4Mul) >> lds.shift4);
                pass |= testBit(sFinal3, (uint32_t)__umul24(x, pfac::kFinal3Mul) >> lds.shiftF3);
                if (HAS_SHORT) pass |= testBit(sShort, x & 0xFFFFu);
                keep = pass != 0;
                hits &= hits - 1;
            }
            const uint64_t keepMask = __ballot(keep);
            if (keepMask) {
                const uint32_t at = qn + laneRankIn(keepMask);
                if (keep) {
                    qPos[at] = (uint16_t)((slot << 10) + (k << 8) + (lane << 2) + (b & 3));
                    qBytes[at] = x;
                }
                qn += (uint32_t)__popcll(keepMask);
            }
            pending = __ballot(hits != 0);
        }
        slot++;

        /* ---- 5. walk when the queue spans the maximum number of tiles */
        if (slot == kMaxSlots) {
            if (qn) drainQueue<MODE, HAS_SHORT>(a, walk, lds, qPos, qBytes, qn, tile0, totalWaves, numDwords);
            qn = 0; slot = 0;
        }
    }
    if (qn) drainQueue<MODE, HAS_SHORT>(a, walk, lds, qPos, qBytes, qn, tile0, totalWaves, numDwords);
}

/* ---------------------------------------------------------- naive kernel */

/* One thread per input byte, no prefilter: the reference's algorithm with
 * only the initial-state row staged in LDS.  Alignment-agnostic. */
template <int MODE>
__global__ __launch_bounds__(256) void pfac_scan_naive(ScanArgs a)
{
    __shared__ int sInit[pfac::kCharSet];
    sInit[threadIdx.x] = a.initialRow[threadIdx.x];
    __syncthreads();
    const Lookup<MODE> lookup(a);
    const size_t n = a.n;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) {
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
        a.out[j] = match;
    }
}

/* ------------------------------------------------------------- launching */

size_t filterLdsBytes(const PFAC_context *c)
{
    size_t bytes = ((size_t(1) << c->filter.log2Bits) + (size_t(1) << c->filter.log2Bits4) +
                    (size_t(1) << c->filter.log2BitsF3)) / 8;
    if (c->filter.hasShort) bytes += 65536 / 8;
    bytes += pfac::kCharSet * sizeof(int);
    if (c->perfMode == PFAC_SPACE_DRIVEN) bytes += pfac::kCharSet * sizeof(Int2);
    bytes += (size_t)kWavesPerBlock * kQueueCap * (sizeof(uint16_t) + sizeof(uint32_t));
    return bytes;
}

template <int MODE, bool HAS_SHORT>
hipError_t launchFilter(const PFAC_context *c, const ScanArgs &a)
{
    auto kernel = pfac_scan_filter<MODE, HAS_SHORT>;
    const size_t lds = filterLdsBytes(c);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int perCU = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, kBlockThreads, lds);
    if (e != hipSuccess) return e;
    if (perCU < 1) perCU = 1;
    const size_t numTiles = (a.n + kTileBytes - 1) / kTileBytes;
    size_t blocks = (numTiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const size_t resident = (size_t)(c->multiProcessorCount > 0 ? c->multiProcessorCount : 256) * perCU;
    if (blocks > resident) blocks = resident;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlockThreads), lds, 0, a);
    return hipGetLastError();
}

template <int MODE>
hipError_t launchNaive(const PFAC_context *c, const ScanArgs &a)
{
    size_t blocks = (a.n + 255) / 256;
    const size_t cap = (size_t)(c->multiProcessorCount > 0 ? c->multiProcessorCount : 256) * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(pfac_scan_naive<MODE>, dim3((unsigned)blocks), dim3(256), 0, 0, a);
    return hipGetLastError();
}

template <int MODE>
hipError_t launchMode(const PFAC_context *c, const ScanArgs &a, bool vectorOk)
{
    if (c->kernelVariant == PFACX_KERNEL_NAIVE || !vectorOk) return launchNaive<MODE>(c, a);
    return c->filter.hasShort ? launchFilter<MODE, true>(c, a) : launchFilter<MODE, false>(c, a);
}

uint32_t clampExtent(size_t bytes) { return bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bytes; }

PFAC_status_t scan(PFAC_handle_t handle, char *d_input_string, size_t input_size, int *d_matched_result, bool hashed)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    const PFAC_context *c = handle;
    if (!c->d_initialRow || !c->d_gram3 || !c->d_gram4 || !c->d_final3 || !c->d_shortBits) return PFAC_STATUS_INTERNAL_ERROR;
    if (hashed ? (!c->d_hashRow || !c->d_hashVal || !c->d_hashFat || !c->d_initialRowInfo) : !c->d_dense)
        return PFAC_STATUS_INTERNAL_ERROR;

    ScanArgs a{};
    a.in = reinterpret_cast<const unsigned char *>(d_input_string);
    a.out = d_matched_result;
    a.n = input_size;
    a.dense = c->d_dense;
    a.hashRow = c->d_hashRow;
    a.hashVal = c->d_hashVal;
    a.denseBytes = clampExtent(c->h_dense.size() * sizeof(int));
    a.hashRowBytes = clampExtent(c->h_hashRow.size() * sizeof(Int2));
    a.hashValBytes = clampExtent(c->h_hashVal.size() * sizeof(Int2));
    a.hashFat = reinterpret_cast<const i32x4 *>(c->d_hashFat);
    a.hashFatBytes = clampExtent(c->h_hashVal.size() * sizeof(pfac::Int4));
    a.initialRowInfo = c->d_initialRowInfo;
    a.initialRow = c->d_initialRow;
    a.gram3 = c->d_gram3;
    a.shortBits = c->d_shortBits;
    a.gram4 = c->d_gram4;
    a.final3 = c->d_final3;
    a.log2Bits = c->filter.log2Bits;
    a.log2Bits4 = c->filter.log2Bits4;
    a.log2BitsF3 = c->filter.log2BitsF3;
    a.numFinal = c->fa.numPatterns;
    a.initialState = c->fa.initialState;

    /* the buffer-resource ("texture") path addresses the table with 32-bit byte
     * offsets; the reference fails the texture bind for an oversized table the
     * same way (PFAC_kernel.cu:139-142) */
    const bool tex = (c->textureMode == PFAC_TEXTURE_ON);
    if (tex) {
        const size_t biggest = hashed ? c->h_hashVal.size() * sizeof(pfac::Int4) : c->h_dense.size() * sizeof(int);
        if (biggest > 0xFFFFFFFFull) return PFAC_STATUS_CUDA_ALLOC_FAILED;
    }
    const bool vectorOk = ((reinterpret_cast<uintptr_t>(a.in) & 3u) == 0) &&
                          ((reinterpret_cast<uintptr_t>(a.out) & 15u) == 0);
    hipError_t e;
    if (hashed) e = tex ? launchMode<HASH_BUFFER>(c, a, vectorOk) : launchMode<HASH_GLOBAL>(c, a, vectorOk);
    else        e = tex ? launchMode<DENSE_BUFFER>(c, a, vectorOk) : launchMode<DENSE_GLOBAL>(c, a, vectorOk);
    return e == hipSuccess ? PFAC_STATUS_SUCCESS : PFAC_STATUS_INTERNAL_ERROR;
}

} // namespace

extern "C" {

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

/* Compacted output is SURVEY.md section 8(f) rank 1 ("next"); not built yet. */
PFAC_status_t PFAC_reduce_kernel(PFAC_handle_t, int *, int, int *, int *, int *, int *, int *)
{
    return PFAC_STATUS_INTERNAL_ERROR;
}

PFAC_status_t PFAC_reduce_inplace_kernel(PFAC_handle_t, int *, int, int *, int *, int *, int *, int *)
{
    return PFAC_STATUS_INTERNAL_ERROR;
}

} /* extern "C" */

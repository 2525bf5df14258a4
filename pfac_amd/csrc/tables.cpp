/*
 * tables.cpp -- materialise the two transition-table layouts from the trie.
 *
 *  dense  ("time-driven"):  int[numStates][256], TRAP (-1) filled
 *                           ref PFAC_create2DTable, PFAC/src/PFAC.cpp:345-402
 *  hashed ("space-driven"): rowPtr int2[numStates] = {offset, (k<<16)|(S-1)}
 *                           valPtr int2[sum S]     = {nextState, ch}, empty = {-1,-1}
 *                           ref PFAC_createHashTable, PFAC/src/PFAC.cpp:422-562
 *
 * The hashed layout is reproduced exactly (bucket ladder, smallest
 * collision-free multiplier, running offsets in state order) so that the
 * device tables are byte-identical to what the reference would upload; the
 * parity tests compare them with the oracle's.
 */
#include <cstring>
#include <new>

#include "pfac_host.h"

namespace pfac {

PFAC_status_t buildDenseTable(const Automaton &fa, std::vector<int> &dense)
{
    try {
        dense.assign((size_t)fa.numStates * kCharSet, kTrapState);
    } catch (const std::bad_alloc &) {
        dense.clear();
        return PFAC_STATUS_ALLOC_FAILED;
    }
    for (int s = 0; s < fa.numStates; s++) {
        int *row = dense.data() + (size_t)s * kCharSet;
        for (int e = fa.edgeBegin[s]; e < fa.edgeBegin[s + 1]; e++) row[fa.edgeCh[e]] = fa.edgeNext[e];
    }
    return PFAC_STATUS_SUCCESS;
}

/* bucket count for a state with `fanout` valid transitions (ref PFAC.cpp:449-473) */
static int bucketCount(int fanout)
{
    if (fanout == 0) return 0;
    if (fanout == 1) return 1;
    if (fanout == 2) return 4;
    if (fanout <= 4) return 16;
    if (fanout == 5) return 32;
    if (fanout <= 8) return 64;
    if (fanout <= 11) return 128;
    if (fanout <= 255) return 256;
    return -1;
}

static inline int slotOf(int k, int ch, int buckets) { return ((k * ch) % kHashP) % buckets; }

PFAC_status_t buildHashTable(const Automaton &fa, std::vector<Int2> &rowPtr,
                             std::vector<Int2> &valPtr)
{
    const int S = fa.numStates;
    try {
        rowPtr.assign((size_t)S, Int2{-1, -1});
    } catch (const std::bad_alloc &) { return PFAC_STATUS_ALLOC_FAILED; }

    size_t total = 0;
    for (int s = 0; s < S; s++) {
        const int fan = fa.edgeBegin[s + 1] - fa.edgeBegin[s];
        const int buckets = bucketCount(fan);
        if (buckets < 0) { rowPtr.clear(); return PFAC_STATUS_INTERNAL_ERROR; }
        if (buckets == 0) continue;
        rowPtr[s].x = (int)total;
        rowPtr[s].y = buckets - 1;
        total += (size_t)buckets;
    }
    try {
        valPtr.assign(total, Int2{-1, -1});        /* ref memset 0xFF, PFAC.cpp:496 */
    } catch (const std::bad_alloc &) { rowPtr.clear(); return PFAC_STATUS_ALLOC_FAILED; }

    for (int s = 0; s < S; s++) {
        const int b = fa.edgeBegin[s], e = fa.edgeBegin[s + 1];
        if (b == e) continue;
        const int buckets = rowPtr[s].y + 1;
        int k = -1;
        if (buckets == 1 || buckets == 256) {
            k = 1;                                 /* ref PFAC.cpp:506-519 */
        } else {
            /* smallest k in [1,256] whose slots are pairwise distinct (ref :520-542) */
            for (int cand = 1; cand <= 256 && k < 0; cand++) {
                uint64_t used[4] = {0, 0, 0, 0};
                bool ok = true;
                for (int i = b; i < e; i++) {
                    const int slot = slotOf(cand, fa.edgeCh[i], buckets);
                    const uint64_t bit = uint64_t(1) << (slot & 63);
                    if (used[slot >> 6] & bit) { ok = false; break; }
                    used[slot >> 6] |= bit;
                }
                if (ok) k = cand;
            }
            if (k < 0) { rowPtr.clear(); valPtr.clear(); return PFAC_STATUS_INTERNAL_ERROR; }
        }
        Int2 *bucket = valPtr.data() + rowPtr[s].x;
        for (int i = b; i < e; i++) {
            Int2 &v = bucket[slotOf(k, fa.edgeCh[i], buckets)];
            v.x = fa.edgeNext[i];
            v.y = fa.edgeCh[i];
        }
        rowPtr[s].y |= (k << 16);                  /* HASH_KEY_K_MASKBITS, PFAC_P.h:89-91 */
    }
    return PFAC_STATUS_SUCCESS;
}

/*
 * Device-only "chained" transition table for the gfx950 walker (scan_*.hip), used in BOTH perf modes.
 *
 * The reference's hashed layout (above) answers one transition with two dependent loads (rowPtr[state], then
 * valPtr[...], PFAC_kernel_spaceDriven.cu:76-124) and gives every state a bucket.  The walker's table is hashed per state
 * like the reference's -- a power-of-two bucket, the smallest multiplier k without a collision -- with a cheaper family,
 * slot = ((k * ch) >> 7) & (S - 1) (pfac_context.h: chainSlotOf), and
 *   - a slot carries the bucket {offset, k, S} of the state the walker will be in next, so a transition needs ONE
 *     dependent 16-byte load;
 *   - a slot carries the single-successor chain that follows `next`: while the current state is not final and has
 *     exactly one outgoing transition, the byte of that transition is appended (up to kChainMax bytes; up to kChainMaxWide
 *     in a WIDE bucket, whose slots keep the bytes behind the eighth in their extension units: pfac_context.h) and the state
 *     advances.  The walker compares the chain against the input and lands directly in the end state.  Skipping is
 *     exact: the skipped states are not final, so they could not have changed the reported match, and a mismatch
 *     anywhere in the chain is the trap state (PFAC_CPU.cpp:76-96);
 *   - only states a walk can LAND in have a bucket (nine out of ten states of a Snort-scale set are inside chains),
 *     buckets are as small as the hash family allows (the smallest power of two >= the fan-out that has a
 *     collision-free k <= 255; the reference's ladder gives a state with nine successors 128 slots) and they are laid out
 *     breadth first, the top of the trie -- where walks spend their time -- in one contiguous piece: the table of the
 *     30 k-pattern bench set shrinks from 11.9 MB to a few MB that stay in the 4 MiB L2 of an XCD next to the stream.
 * The end state's number is only needed when it is final (it is the pattern ID).  A final leaf keeps it in endRow (a
 * leaf has no bucket); a final state with successors (a pattern that is a prefix of another) keeps it in chain[4..7],
 * and such a slot's chain is cut to <= 3 bytes (the cut lands on a non-final chain state; the next slot carries on
 * from there).
 *
 * The array is N slot headers followed by N EXTENSION UNITS, unit i (at N + i) belonging to slot i: a slot of a wide bucket
 * keeps its chain bytes 8..22 there (pfac_context.h).  The headers lie exactly where the table of rounds 2-4 had them --
 * what a walk touches while it follows no long single-successor run is unchanged --, and a unit is only ever fetched for
 * a long slot (or, by a walker that has found its stream to be full of near misses, together with the header).
 * Behind the buckets the headers carry two more regions:
 *   [rootRow, rootRow + 256)   the bucket of the initial state, indexed by the byte itself (hash k = 128, S = 256);
 *   [jumpBase + 2^J, jumpBase + 2 * 2^J) the LONG jump table: the same slots with chains of up to kChainMaxWide bytes;
 *   [jumpBase, jumpBase + 2^J) the JUMP table: one slot per 4-byte pattern prefix whose first three states are not
 *                              final, at hash(prefix), encoded as a transition on the first byte with the other
 *                              three as the head of its chain.  A walk starts there -- the prefilter has just
 *                              established that the four bytes are (probably) a prefix -- and lands four or
 *                              more bytes deep with its first gathered load.  The table is an accelerator, not
 *                              an index: prefixes that collide (first come, first served), that pass a final
 *                              state, or that the prefilter let through wrongly simply are not there, and the
 *                              walker falls back to the initial state's bucket.
 */
namespace {

#ifndef PFAC_WIDE_BUCKETS
#define PFAC_WIDE_BUCKETS 1                    /* 0: no wide buckets (every chain <= kChainMax: the table of rounds 2-4) */
#endif

/* a slot as the builder makes it: the 16-byte header and the chain bytes 8..22 that go to the slot's extension unit */
struct BuiltSlot {
    ChainSlot hdr;
    unsigned char ext[16];
};

struct ChainBuilder {
    const Automaton &fa;
    std::vector<ChainSlot> &slots;                 /* the buckets; root row and jump table are appended at the end */
    std::vector<ChainSlot> units;                  /* the extension unit of every slot of `slots` (zeros but for the long slots) */
    std::vector<int> bucketOff;                    /* per state: first slot of its bucket, -1 = none yet */
    std::vector<uint32_t> bucketKS;                /* per state: wide << 16 | k << 8 | (S - 1) */
    static int chainSlotIn(int k, int ch, int S) { return ((k * ch) >> 7) & (S - 1); }     /* pfac::chainSlotOf */
    std::vector<int> pending;                      /* states whose bucket is allocated but not filled, in allocation order */
    bool failed = false;
    size_t wideBuckets = 0, longSlots = 0;

    bool wideBuckets_ = PFAC_WIDE_BUCKETS != 0;    /* false: every chain <= kChainMax (the table of rounds 2-4; the NARROW table of round 6) */

    ChainBuilder(const Automaton &a, std::vector<ChainSlot> &out, bool wideOn)
        : fa(a), slots(out), bucketOff((size_t)a.numStates, -1), bucketKS((size_t)a.numStates, 0u), wideBuckets_(wideOn && PFAC_WIDE_BUCKETS != 0) {}

    static ChainSlot emptySlot()
    {
        ChainSlot s;
        std::memset(&s, 0, sizeof(s));
        s.meta = kSlotEmpty;
        s.endRow = -1;
        return s;
    }
    static ChainSlot zeroUnit()
    {
        ChainSlot s;
        std::memset(&s, 0, sizeof(s));
        return s;
    }
    int fanout(int s) const { return fa.edgeBegin[s + 1] - fa.edgeBegin[s]; }

    /* single-successor, non-final states in a row from `state` on (what a slot into `state` can fold), up to `limit` */
    int naturalChain(int state, int limit) const
    {
        int k = 0;
        while (k < limit && state > fa.numPatterns && fanout(state) == 1) {
            state = fa.edgeNext[fa.edgeBegin[state]];
            k++;
        }
        return k;
    }

    /* the bucket of a state with successors: allocated on first use.  WIDE (its slots may fold up to kChainMaxWide chain
     * bytes, those behind the eighth in the slot's extension unit) if one of its transitions is followed by more
     * single-successor bytes than a header holds: a walk through such a run then takes 24 bytes per step instead of 8. */
    void needBucket(int state)
    {
        if (bucketOff[state] >= 0) return;
        const int b = fa.edgeBegin[state], e = fa.edgeBegin[state + 1], fan = e - b;
        int S = 1;
        while (S < fan) S *= 2;
        int k = -1;
        for (; S <= 256 && k < 0; S *= 2) {
            for (int cand = 1; cand <= 255 && k < 0; cand++) {
                uint64_t used[4] = {0, 0, 0, 0};
                bool ok = true;
                for (int i = b; i < e && ok; i++) {
                    const int slot = chainSlotIn(cand, fa.edgeCh[i], S);
                    const uint64_t bit = uint64_t(1) << (slot & 63);
                    ok = !(used[slot >> 6] & bit);
                    used[slot >> 6] |= bit;
                }
                if (ok) k = cand;
            }
            if (k >= 0) break;
        }
        if (k < 0) { failed = true; return; }      /* cannot happen: k = 128, S = 256 is the identity */
        bool wide = false;
        if (wideBuckets_)
            for (int i = b; i < e && !wide; i++) wide = naturalChain(fa.edgeNext[i], kChainMax + 1) > kChainMax;
        bucketOff[state] = (int)slots.size();
        bucketKS[state] = ((uint32_t)wide << 16) | ((uint32_t)k << 8) | (uint32_t)(S - 1);
        slots.resize(slots.size() + (size_t)S, emptySlot());
        units.resize(slots.size(), zeroUnit());
        if (wide) wideBuckets++;
        pending.push_back(state);
    }

    /* the slot of the transition on byte ch into state `next`, folding at most `cap` chain bytes (kChainMax, or kChainMaxWide
     * for a slot of a wide bucket); forced: the first chain bytes are given (a path through non-final states that may
     * branch: the jump slots) and `next` is the state behind them */
    BuiltSlot makeSlot(int ch, int next, int cap, const unsigned char *forced = nullptr, int numForced = 0)
    {
        BuiltSlot out;
        out.hdr = emptySlot();
        std::memset(out.ext, 0, sizeof(out.ext));
        ChainSlot &s = out.hdr;
        if (next < 0) return out;
        unsigned char chain[kChainMaxWide + 1];
        auto follow = [&](int limit, int &end) {
            int k = 0;
            std::memset(chain, 0, sizeof(chain));
            for (; k < numForced; k++) chain[k] = forced[k];
            end = next;
            while (k < limit && end > fa.numPatterns && fanout(end) == 1) {
                chain[k++] = fa.edgeCh[fa.edgeBegin[end]];
                end = fa.edgeNext[fa.edgeBegin[end]];
            }
            return k;
        };
        int cur;
        int k = follow(cap, cur);
        if (cur <= fa.numPatterns && fanout(cur) > 0 && k > 3)          /* final with successors: the ID needs chain[4..7]: the chain is cut so that the */
            k = follow(k >= 8 ? k - 4 : 3, cur);                         /* NEXT slot reaches that state with <= 3 chain bytes (the cut lands on a non-final chain state) */
        const bool leaf = fanout(cur) == 0;
        const bool fin = cur <= fa.numPatterns;
        if (!leaf) needBucket(cur);
        const uint32_t hashK = leaf ? 0u : (bucketKS[cur] >> 8) & 0xFFu;        /* 1..255; 0 = leaf */
        const uint32_t sizeMask = leaf ? 0u : bucketKS[cur] & 0xFFu;             /* S-1 <= 255 */
        const bool wideEnd = !leaf && (bucketKS[cur] >> 16) != 0;
        s.meta = (uint32_t)ch | ((uint32_t)k << kSlotLenShift) | (fin ? kSlotFinal : 0u) | (wideEnd ? kSlotWide : 0u) | (hashK << 16) | (sizeMask << 24);
        s.endRow = leaf ? (fin ? cur : -1) : bucketOff[cur];
        std::memcpy(s.chain, chain, 8);
        if (k > 8) std::memcpy(out.ext, chain + 8, (size_t)(k - 8));
        if (fin && !leaf) std::memcpy(s.chain + 4, &cur, sizeof(int));           /* k <= 3 here */
        if (k > kChainMax) longSlots++;
        return out;
    }

    /* fill the buckets allocated so far, and those their slots allocate, breadth first */
    void drain()
    {
        for (size_t at = 0; at < pending.size() && !failed; at++) {
            const int state = pending[at];
            const int k = (int)((bucketKS[state] >> 8) & 0xFFu), S = (int)(bucketKS[state] & 0xFFu) + 1;
            const bool wide = (bucketKS[state] >> 16) != 0;
            for (int e = fa.edgeBegin[state]; e < fa.edgeBegin[state + 1]; e++) {
                const BuiltSlot slot = makeSlot(fa.edgeCh[e], fa.edgeNext[e], wide ? kChainMaxWide : kChainMax);   /* may grow `slots` */
                const size_t at0 = (size_t)bucketOff[state] + (size_t)chainSlotIn(k, fa.edgeCh[e], S);
                slots[at0] = slot.hdr;
                std::memcpy(&units[at0], slot.ext, sizeof(slot.ext));
            }
        }
        pending.clear();
    }
};

} // namespace

/* narrow = true (round 6): the same table without wide buckets, long jump table and extension units -- N' = buckets + 256 + 2^J headers and nothing
 * else.  The tiled kernel walks it on streams that are not full of near misses (calls below 32 MiB, pattern-dense chunks): a long slot's unit is
 * fetched on the spot there, the whole wave waiting, and on text nearly every step of a wave's 256 walks had one or two lanes at such a slot
 * (C3 through the tiled kernel alone: 704-757 GB/s before the wide buckets, 652-680 with them). */
PFAC_status_t buildChainedHashTable(const Automaton &fa, std::vector<ChainSlot> &slots, int &jumpLog2, bool narrow)
{
    const int F = fa.numPatterns, init = fa.initialState;
    /* 4-byte prefixes: {key, state at depth 4} */
    struct Prefix { uint32_t key; int state; };
    std::vector<Prefix> prefixes;
    try {
        slots.clear();
        if (fa.numStates <= init) {                                /* no patterns: root row + smallest jump table, all empty */
            jumpLog2 = kJumpLog2Min;
            if (narrow) { slots.assign((size_t)kCharSet + (size_t(1) << jumpLog2), ChainBuilder::emptySlot()); return PFAC_STATUS_SUCCESS; }
            slots.assign((size_t)kCharSet + (size_t(2) << jumpLog2), ChainBuilder::emptySlot());
            slots.resize(2 * slots.size(), ChainBuilder::zeroUnit());
            return PFAC_STATUS_SUCCESS;
        }
        for (int e1 = fa.edgeBegin[init]; e1 < fa.edgeBegin[init + 1]; e1++) {
            const int s1 = fa.edgeNext[e1];
            if (s1 <= F) continue;
            for (int e2 = fa.edgeBegin[s1]; e2 < fa.edgeBegin[s1 + 1]; e2++) {
                const int s2 = fa.edgeNext[e2];
                if (s2 <= F) continue;
                for (int e3 = fa.edgeBegin[s2]; e3 < fa.edgeBegin[s2 + 1]; e3++) {
                    const int s3 = fa.edgeNext[e3];
                    if (s3 <= F) continue;
                    for (int e4 = fa.edgeBegin[s3]; e4 < fa.edgeBegin[s3 + 1]; e4++)
                        prefixes.push_back({(uint32_t)fa.edgeCh[e1] | ((uint32_t)fa.edgeCh[e2] << 8) | ((uint32_t)fa.edgeCh[e3] << 16) |
                                                ((uint32_t)fa.edgeCh[e4] << 24),
                                            fa.edgeNext[e4]});
                }
            }
        }
        jumpLog2 = kJumpLog2Min;
#ifndef PFAC_JUMP_SPARSITY
#define PFAC_JUMP_SPARSITY 8                   /* jump slots per 4-byte pattern prefix (a prefix that finds its slot taken walks from the initial state) */
#endif
        while (jumpLog2 < kJumpLog2Max && (size_t(1) << jumpLog2) < (size_t)PFAC_JUMP_SPARSITY * prefixes.size()) jumpLog2++;
        ChainBuilder b(fa, slots, !narrow);
        std::vector<ChainSlot> root((size_t)kCharSet, ChainBuilder::emptySlot()), jump(size_t(1) << jumpLog2, ChainBuilder::emptySlot());
        /* the LONG jump table: the same prefixes in the same places, but a slot folds up to kChainMaxWide bytes (chain bytes 8.. in
         * its extension unit): the start of a walker that expects long single-successor runs (scan_common.h: StageLane) */
        std::vector<ChainSlot> jumpLong(jump.size(), ChainBuilder::emptySlot()), jumpLongUnits(jump.size(), ChainBuilder::zeroUnit());
        /* the top of the trie first: what the initial state's transitions land in, then what the jump slots land in,
         * then everything below, level by level */
        for (int e = fa.edgeBegin[init]; e < fa.edgeBegin[init + 1]; e++) root[fa.edgeCh[e]] = b.makeSlot(fa.edgeCh[e], fa.edgeNext[e], kChainMax).hdr;
        for (const Prefix &p : prefixes) {
            ChainSlot &dst = jump[jumpHash(p.key, jumpLog2)];
            if (!(dst.meta & kSlotEmpty)) continue;            /* taken: this prefix walks from the initial state */
            const unsigned char rest[3] = {(unsigned char)(p.key >> 8), (unsigned char)(p.key >> 16), (unsigned char)(p.key >> 24)};
            dst = b.makeSlot((int)(p.key & 0xFFu), p.state, kChainMax, rest, 3).hdr;
            if (narrow) continue;
            const BuiltSlot wide = b.makeSlot((int)(p.key & 0xFFu), p.state, PFAC_WIDE_BUCKETS ? kChainMaxWide : kChainMax, rest, 3);
            jumpLong[jumpHash(p.key, jumpLog2)] = wide.hdr;
            std::memcpy(&jumpLongUnits[jumpHash(p.key, jumpLog2)], wide.ext, sizeof(wide.ext));
        }
        b.drain();
        if (b.failed) return PFAC_STATUS_INTERNAL_ERROR;
        slots.insert(slots.end(), root.begin(), root.end());
        slots.insert(slots.end(), jump.begin(), jump.end());
        if (narrow) return PFAC_STATUS_SUCCESS;
        /* the extension units, slot for slot: unit i belongs to slot i (root row and the short jump table never have one) */
        b.units.resize(slots.size(), ChainBuilder::zeroUnit());
        slots.insert(slots.end(), jumpLong.begin(), jumpLong.end());
        b.units.insert(b.units.end(), jumpLongUnits.begin(), jumpLongUnits.end());
        slots.insert(slots.end(), b.units.begin(), b.units.end());
    } catch (const std::bad_alloc &) { return PFAC_STATUS_ALLOC_FAILED; }
    return PFAC_STATUS_SUCCESS;
}

} // namespace pfac

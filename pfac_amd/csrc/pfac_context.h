/*
 * pfac_context.h -- private state behind PFAC_handle_t, shared by libpfac.so
 * (host side) and the kernel module libpfac_gfx950.so.
 *
 * Plays the role of the reference's struct PFAC_context
 * (PFAC/include/PFAC_P.h:94-178) but is laid out for this implementation:
 * the automaton is kept as a CSR edge list in insertion order (the reference
 * keeps vector<vector<TableEle>>), and the context additionally owns the
 * LDS prefilter bitmaps that only this implementation has.
 */
#ifndef PFAC_CONTEXT_H_
#define PFAC_CONTEXT_H_

#include <stddef.h>
#include <stdint.h>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <utility>
#include <vector>

#include "PFAC.h"
#include "pfac_ext.h"
#include "pfac_module.h"

namespace pfac {

constexpr int kCharSet = 256;                 /* ref CHAR_SET, PFAC_P.h:181 */
constexpr int kTrapState = -1;                /* ref TRAP_STATE 0xFFFFFFFF, PFAC_P.h:182 */
constexpr int kHashP = 257;                   /* ref hash_p, PFAC.cpp:439 */
constexpr size_t kTexMaxEntries = size_t(1) << 27;   /* ref MAXIMUM_WIDTH_1DTEX, PFAC.cpp:69 */
constexpr int kFileNameLen = 256;             /* ref FILENAME_LEN, PFAC_P.h:34 */

struct Int2 { int x, y; };                    /* device layout of the hashed tables (CUDA int2) */
#ifndef PFAC_WORK_PARTS
#define PFAC_WORK_PARTS 2
#endif
constexpr int kWorkParts = PFAC_WORK_PARTS;                /* the scan kernel hands out chunks in order within each of these input parts */
constexpr int kWorkCounterWords = 64 * 32 + 64;   /* 64 counters, one per 128-byte line (0..31 the parts of the input; 32..34, 48: see below), + the launch statistics */
constexpr int kDenseCountWord = 32 * 32;         /* line 32 / 34: dense chunks listed by a filter launch, for the simple kernel behind it; launches alternate */
constexpr int kDenseCountWordB = 34 * 32;
constexpr int kDoneWord = 33 * 32;               /* blocks of the running filter launch that have finished: the last one publishes the statistics and leaves
                                                    every counter zero for the next launch (no memset in front of a launch) */
constexpr int kModeHintWord = 35 * 32;           /* full-result filter kernel: 1 = most scanning waves of the last launch ended it in stage mode (near-miss stream): the next
                                                    launch on this handle starts there (scan_common.h: StageLane); kModeVotesWord counts them during a launch */
constexpr int kModeVotesWord = 36 * 32;
constexpr int kTiledDenseWord = 37 * 32;         /* tiled kernel scanning a whole big call: groups it walked in dense mode, groups in all, waves that are through (three words);
                                                    the last wave out tells the host whether the stream is pattern-dense (hostHint[1]) and leaves them zero */
constexpr int kHostPairCountWord = 4;             /* word of the handle's mapped host memory (h_modeHint) the first ordering launch (pfac_order_count) writes the number of pairs of a compacted-output call to */
constexpr int kStatsPublishedWord = 48 * 32;     /* 64-bit: the kStatsCount statistics of the last finished filter launch, then its dense chunks */
constexpr int kStatsWord = 64 * 32;              /* 64-bit launch statistics of the scan kernel live here, behind the part counters (PFACX_getScanStats) */
constexpr int kStatsCount = 6;                  /* walker rounds, lane steps, walks started, level-1 hits, positions scanned, ladder candidates */
/* shape of the scan kernel (scan_*.hip), reported by PFACX_getScanStats */
#ifndef PFAC_WALK_SETS
#define PFAC_WALK_SETS 2                       /* independent walks per lane, compacted-output kernel (every candidate behind the level-4 test walks there) */
#endif
#ifndef PFAC_WALK_SETS_FULL
#define PFAC_WALK_SETS_FULL 1                  /* ... full-result kernel: behind the prefix ladder one walk per lane keeps up, and is 1 % (C3) to 5 % (C5) faster */
#endif
constexpr int kChunkTiles = 2;                 /* KiB of input a wave stages at a time */
constexpr int kChainMax = 7;                  /* bytes of single-successor chain a slot header folds in ...                                */
constexpr int kChainMaxWide = 23;             /* ... a slot of a WIDE bucket: 8 in the header + 15 in its extension unit                    */
/* 16-byte device slot of the chained hashed table: one gathered 16-byte load per transition.
 * meta = edge byte | chain length << 8 (5 bits) | flags | k << 16 | (S-1) << 24, where {k, S} are the hash
 * parameters of the END state's bucket: the successor on byte ch sits in slot ((k * ch) >> 7) & (S-1) of it
 * (1 <= k <= 255, S a power of two <= 256, k the smallest multiplier without a collision; k = 128, S = 256 is the
 * identity and always works; k = 0: the end state is a LEAF, it has no bucket).  The reference's hashed layout uses
 * ((k * ch) mod 257) & (S-1) (PFAC.cpp:506-542); this table is device-only and its own: the multiply-shift costs a
 * walk step four instructions instead of nine.
 * WIDE buckets (round 5: long single-successor runs -- near misses of long patterns, BASELINE config 5 -- were eight
 * dependent steps of 8 bytes): a slot of a wide bucket folds up to kChainMaxWide chain bytes, bytes 0..7 in the header and
 * bytes 8..22 in the slot's EXTENSION UNIT.  The table is N headers followed by N units, the unit of slot i at N + i, so
 * a walker that expects long slots (kSlotWide of the slot that led into the bucket) can fetch header and unit as two
 * INDEPENDENT loads and take 24 bytes with one dependent round trip.  Only slots of wide buckets have chains longer
 * than kChainMax; the initial state's bucket and the jump table are never wide. */
struct ChainSlot {
    uint32_t meta;
    int endRow;                               /* hashRowPtr[end].x (first slot of the end state's bucket);
                                                 if the end state is a final LEAF: its pattern ID        */
    unsigned char chain[8];                   /* chain bytes 0..7, zero padded; if the end state is final and
                                                 has successors: <= 3 chain bytes, pattern ID in [4..7]  */
};
constexpr size_t kGram3LdsBytes = 32 * 1024;     /* LDS set aside for the level-1 bitmap (2^18 bits at most): the ladder behind it sits at a compile-time address */
constexpr size_t kFilterLdsBudget = 97 * 1024;   /* LDS bytes the prefilter bitmaps may take together (pattern_compiler.cpp); the rest of the
                                                    CU's 160 KiB is the scanning waves' queues and stages (scan_*.hip checks the sum) */
constexpr uint32_t kSlotLenShift = 8, kSlotLenMask = 0x1Fu;   /* chain length: bits 8..12 */
constexpr uint32_t kSlotFinal = 1u << 13;     /* the end state is a final state                         */
constexpr uint32_t kSlotEmpty = 1u << 14;     /* no transition in this slot                             */
constexpr uint32_t kSlotWide = 1u << 15;      /* the end state's bucket is wide: its slots may be long (chain bytes 8.. in their extension units) */
constexpr uint32_t kSlotKMask = 0xFFu << 16;  /* k == 0: the end state has no outgoing transition (a leaf) */
static_assert(sizeof(ChainSlot) == 16, "ChainSlot is read as one 16-byte load");
static_assert(kChainMaxWide <= (int)kSlotLenMask && kChainMaxWide == 8 + 15, "the chain length has five bits; a unit holds chain bytes 8..22 and the byte behind the chain must still be one of 24");
constexpr uint32_t kChainRootMeta = (128u << 16) | (255u << 24);   /* hash parameters of the initial state's bucket: slot of byte b = b */
inline bool chainSlotLeaf(uint32_t meta) { return (meta & kSlotKMask) == 0; }
inline uint32_t chainSlotOf(uint32_t meta, uint32_t ch) { return ((((meta >> 16) & 0xFFu) * ch) >> 7) & (meta >> 24); }

/* One compiled pattern set: patterns + trie.  Independent of perfMode. */
struct Automaton {
    std::vector<unsigned char> file;          /* raw pattern-file bytes                          */
    int numPatterns = 0;                      /* F                                               */
    std::vector<int> patternOff;              /* [F+1] by ID, byte offset into file              */
    std::vector<int> patternLen;              /* [F+1] by ID ([0] = 0)                           */
    std::vector<int> sortedId;                /* [F] IDs in (signed-char, prefix-first) order    */
    int maxPatternLen = 0;
    size_t trailingBytes = 0;                 /* bytes behind the last '\n' of the pattern file: ignored, like the reference ignores them */
    int initialState = 0;                     /* F+1                                             */
    int numStates = 0;                        /* next unused id; counts unused state 0           */
    int numLeaves = 0;
    /* CSR over states; edges of a state are in insertion order */
    std::vector<int> edgeBegin;               /* [numStates+1]                                   */
    std::vector<unsigned char> edgeCh;
    std::vector<int> edgeNext;
};

/* Prefilter.  Level 1, tested for every input position: a start position can only produce a
 * non-zero result if its first three bytes hit gram3 (patterns of one or two bytes are folded into it).  Level 2,
 * tested only for the survivors just before they would touch the transition table, is a LADDER of prefix tests over
 * the 20 bytes a candidate brings along: prefix lengths 4, 6, ..., 20.  A trie node at one of these depths is either
 *   S ("stop": walk the table from here) -- a pattern ends at this depth or the next, or at most ladderThin patterns
 *      lie below it, or it is the last level; or
 *   G ("go on": test the next level)      -- everything else; only these have their descendants in the ladder.
 * Both kinds live in ONE Bloom bitmap keyed by a rolling hash of the prefix: S nodes set two bits, G nodes one (two
 * at depth 4, where the test decides which level-1 hits become candidates at all) -- since round 6 all in the ONE dword the hash picks (blocked: a level is one LDS read).  A candidate walks the levels until it hits an S node (-> walk), or neither kind (-> its result is 0).
 * Patterns shorter than 4 bytes bypass the ladder: final3 (length exactly 3) and shortBits (length 1-2).
 * All bitmaps are supersets of the exact sets, so a miss proves the result is 0 (a false positive costs a walk or one
 * more level, never correctness: a matching pattern's nodes are S or G at every level up to its first S node). */
struct Filter {
    int log2Bits = 13;                        /* gram3  */
    int log2BitsLad = 13;                     /* ladder */
    int log2BitsF3 = 10;                      /* final3 */
    bool hasShort = false;
    size_t bitsSet = 0;                       /* population of gram3 */
    size_t bitsSetLad = 0;                    /* population of the ladder bitmap */
    size_t ladderStops = 0, ladderGoOns = 0;  /* S and G nodes inserted */
    int ladderThin = 1;                       /* nodes with at most this many patterns below them are S (raised until the bitmap is sparse enough) */
    uint32_t ladderSalt = 0;                  /* XORed into the depth-4 hash of the ladder (and so into every hash rolled from it): picked by the pattern compiler so that no
                                                 GO-ON node with many patterns below it has its two STOP bits set by its dword's other tenants (pattern_compiler.cpp) */
    int ladderLast = 20;                      /* deepest level of the ladder: kLadderLast, or kLadderDeepLast when the nodes behind kLadderLast fit the bitmap too (few do: most
                                                 paths are alone by then; BASELINE config 5's 24-byte shared prefix is what needs them) */
    /* The tail table (round 5; the veto of the VETO kernels on a ladder stop): a stop node below which ONE pattern is left knows the rest of
     * that pattern.  Entry {tag = the node's ladder hash, hash = the tag rolled on (tailRoll: four bytes a step) over the LAST `bytes` bytes of the pattern (a multiple of four, at most kTailMaxBytes),
     * bytes | depth of the first of them << 8} in slot tailSlot(tag) or tailSlot2(tag); a candidate that stops at such a node rolls
     * its own hash over as many of its bytes and is walked only if the two agree -- a near miss of a long pattern costs a few multiplications
     * instead of a walk through the table.  Only nodes whose hash no other ladder node shares have an entry (a shared hash could veto another
     * pattern's candidate); an entry that finds both its slots taken is left out; no entry = walk, as before. */
    std::vector<uint32_t> tail;               /* 3 words per slot, 2^log2Tail slots; empty: none */
    int log2Tail = 0;
    size_t tailEntries = 0;
    /* ... and its form for a set whose bitmaps leave the LDS no room for it, or whose thin stops outnumber kTailLog2Max slots
     * (Snort-scale: round 6): the same entries in DEVICE memory, consulted by the VETO = 2 kernels with one gathered 16-byte load per
     * candidate that was told to stop (the lines a near-miss stream asks for stay in L2: it meets the same few hundred stop nodes over
     * and over).  A bucket -- tailGBucket(tag) -- holds two 8-byte entries {tag, (hash & ~kTailGInfoMask) | depth of the first compared
     * byte << 3 | bytes / 4 - 1}: 21 bits of the rolled hash are compared (a near miss that agrees in all of them is walked: harmless),
     * an entry whose bucket is full is left out (no entry = walk).  `from` >= kLadderFirst, so an occupied entry has bits among
     * kTailGFromMask; depths beyond 255 have no entry. */
    std::vector<uint32_t> tailG;              /* 4 words per bucket, 2^log2TailG buckets; empty: none */
    int log2TailG = 0;
    size_t tailGEntries = 0;
    size_t tailCandidates = 0;                /* thin stops that asked for an entry (of either form) */
    int ladderExtend = 0;                     /* ... after this many more levels (0: at once): the deeper test spares the walk of a candidate that
                                                 shares a pattern's prefix up to the thin node and no further */
    /* Skip tags (round 6): the ladder hash of a depth-6 node from which ONE path leads down to depth kLadderLast without a pattern ending on the
     * way and without a thin node (several patterns share at least kLadderLast bytes: BASELINE config 5's 24-byte prefix).  A candidate whose
     * depth-6 hash is such a tag need not be asked about the levels in between: its hash at kLadderLast covers every byte up to there, so whatever
     * does not follow the path dies at that level.  A wave skips a level nobody has to be asked about.  Exact, not a Bloom bit: a tag exists only for a hash
     * no other ladder node shares, so a match -- whose nodes are real nodes -- can only be told to skip by its own path's tag (a stray bit could
     * make a candidate skip the level at which its pattern ends).  The kernels take them as arguments (scalar registers): at most kSkipTagsMax. */
    uint32_t skipTags[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int skipCount = 0;
    std::vector<uint32_t> gram3;              /* 2^log2Bits bits, key c0|c1<<8|c2<<16            */
    std::vector<uint32_t> ladder;             /* 2^log2BitsLad bits                              */
    std::vector<uint32_t> final3;             /* 2^log2BitsF3 bits, 3-byte patterns              */
    std::vector<uint32_t> shortBits;          /* 65536 bits, index c0 | c1<<8                    */
    /* The compacted-output kernel is bound by instruction issue and tests no ladder level behind depth 4, so it has its
     * own pair of bitmaps in the LDS the ladder would take: gram1, a ONE-bit 3-gram filter of kGram1Log2 bits (as few
     * positions pass as through the two-bit gram3 of half the size, for two instructions less per position), and
     * prefix4, the trie's depth-4 nodes alone (two probes; what that kernel's level-4 test needs of the ladder). */
    std::vector<uint32_t> gram1;              /* 2^kGram1Log2 bits */
    std::vector<uint32_t> prefix4;            /* 2^kPrefix4Log2 bits */
};
constexpr int kGram1Log2 = 19, kPrefix4Log2 = 17;
constexpr int kSkipTagsMax = 8, kSkipFromDepth = 6;
constexpr int kDenseFastMaxStates = 8192;      /* 8 MiB of int[S][256]: stays in L2 */

/* the ladder's hash: h(4) = (first four bytes, little endian) * kLadMul0; h(d) = (h(d-2) ^ (bytes d-2, d-1 as a 16-bit
 * little-endian number)) * kLadMul; h(4) also takes the set's salt (ladderStart).  Where a node's bits lie: ladderWord / ladderBit* below (round 6: all in one
 * dword).  scan_filter.hip and tests/filter_model.py evaluate exactly this. */
constexpr int kLadderFirst = 4, kLadderStep = 2, kLadderLast = 20;
constexpr int kLadderDeepLast = 60;            /* a DEEP ladder (Filter::ladderLast) goes on in steps of two down to here: the levels behind kLadderLast are tested by a rolled loop of the
                                                 VETO kernels only; every other kernel walks what is undecided at kLadderLast */
constexpr uint32_t kTailMul = 0x9E3779B1u, kTailMul2 = 0x85EBCA77u;      /* the two slots of a tail entry: (tag * mul) >> (32 - log2Tail) */
constexpr int kTailLog2Max = 12, kTailMinBytes = 6;          /* at most 4096 slots of 12 bytes (LDS); shorter rests are not worth an entry */
#ifndef PFAC_TAIL_MAX_BYTES
#define PFAC_TAIL_MAX_BYTES 16
#endif
constexpr int kTailMaxBytes = PFAC_TAIL_MAX_BYTES;           /* ... and of a longer rest the LAST 16 bytes are compared (any bytes may be: the veto only has to hold for every match;
                                                                near misses differ near the end): the kernel reads them in one go.  (Round 5: 32 -- twice the hash steps for every
                                                                batch of candidates; the model's walk counts are the same to the per cent with 16) */
constexpr uint32_t kTailGInfoMask = 0x7FFu, kTailGFromMask = 0x7F8u;
constexpr int kTailGLog2Min = 8, kTailGLog2Max = 20;
inline uint32_t tailGBucket(uint32_t tag, int log2Buckets) { return (uint32_t)(tag * kTailMul) >> (32 - log2Buckets); }
inline uint32_t tailSlot(uint32_t tag, int log2Slots) { return (uint32_t)(tag * kTailMul) >> (32 - log2Slots); }
inline uint32_t tailSlot2(uint32_t tag, int log2Slots) { return (uint32_t)(tag * kTailMul2) >> (32 - log2Slots); }
constexpr int kLadderLevels = (kLadderLast - kLadderFirst) / kLadderStep + 1;
constexpr uint32_t kLadMul0 = 0x9E3779B1u, kLadMul = 0x85EBCA77u, kLadMulS = 0xC2B2AE3Du, kLadMulG = 0x27D4EB2Fu, kLadMulG2 = 0x165667B1u;
inline uint32_t tailRoll(uint32_t h, uint32_t piece32) { return (h ^ piece32) * kLadMul; }       /* the tail hash: four bytes a step */
inline uint32_t ladderStart(uint32_t first4, uint32_t salt = 0) { return (first4 * kLadMul0) ^ salt; }      /* salt: pfac::Filter::ladderSalt (prefix4: none) */
inline uint32_t ladderRoll(uint32_t h, uint32_t piece16) { return (h ^ piece16) * kLadMul; }
/* Round 6: the ladder bitmap is BLOCKED like level 1 -- all bits of a node lie in ONE dword, so a level costs the kernel one LDS read and no
 * multiplication (rounds 3 - 5: three reads at three hashed places, two 32-bit multiplications): the dword from bits 18.. of h (its byte address
 * is one SDWA AND of h's high half, like level 1's), the bits from four 5-bit fields of h: S = bits [3..7] and [8..12], G = [13..17], the second
 * G bit of depth 4 = [0..4].  (kLadMulS / kLadMulG / kLadMulG2 are still what prefix4 and the layout fingerprint use.) */
#ifndef PFAC_LADDER_BLOCKED
#define PFAC_LADDER_BLOCKED 1                  /* 0: measurement builds of both libraries -- the layout of rounds 3 - 5 (tests/filter_model.py models the blocked one) */
#endif
inline uint32_t ladderWord(uint32_t h, int log2Bits) { return (h >> 18) & ((1u << (log2Bits - 5)) - 1u); }
#if PFAC_LADDER_BLOCKED
inline uint32_t ladderBitS1(uint32_t h, int log2Bits) { return ladderWord(h, log2Bits) * 32u + ((h >> 3) & 31u); }
inline uint32_t ladderBitS2(uint32_t h, int log2Bits) { return ladderWord(h, log2Bits) * 32u + ((h >> 8) & 31u); }
inline uint32_t ladderBitG(uint32_t h, int log2Bits) { return ladderWord(h, log2Bits) * 32u + ((h >> 13) & 31u); }
inline uint32_t ladderBitG2(uint32_t h, int log2Bits) { return ladderWord(h, log2Bits) * 32u + (h & 31u); }
#else
inline uint32_t ladderBitS1(uint32_t h, int log2Bits) { return h >> (32 - log2Bits); }
inline uint32_t ladderBitS2(uint32_t h, int log2Bits) { return (uint32_t)(h * kLadMulS) >> (32 - log2Bits); }
inline uint32_t ladderBitG(uint32_t h, int log2Bits) { return (uint32_t)(h * kLadMulG) >> (32 - log2Bits); }
inline uint32_t ladderBitG2(uint32_t h, int log2Bits) { return (uint32_t)(h * kLadMulG2) >> (32 - log2Bits); }
#endif

constexpr uint32_t kGram3Mul = 0x9A17AFu;     /* 24-bit odd multiplier of the 3-gram hash of gram3, picked by scanning 200 candidates for the lowest false-positive
                                                 rate on the text stream (round 6, for the dword index below: 4.6 % of the Snort-style stream pass level 1 where round 5's
                                                 0x8B92C5 with the product's TOP bits let 4.95 % through; the near-miss stream over the 31 000-pattern set: 6.2 % instead of 8.3 %) */
constexpr uint32_t kGram1Mul = 0x8B92C5u;     /* ... of gram1 (the compacted-output kernel's one-bit level 1): round 4's choice stands there (4.8 % against 5.3 % with the above) */
/* Level 1 is a BLOCKED two-bit Bloom filter: a 3-gram owns two bits of ONE dword, so a position costs the kernel one
 * LDS read (a one-bit filter of twice the size lets slightly fewer positions through, 4.8 % instead of 5.1 % of the
 * Snort-style stream, and leaves the prefix ladder half the LDS).  The dword comes from bits 18.. of the 24 x 24 ->
 * 32 bit product -- the kernels get its BYTE address with ONE instruction, an AND of the product's high half with
 * (words - 1) << 2 (SDWA; the top bits, round 5, needed a shift and an AND: one instruction of nine per position) --, the first
 * bit from the low five bits of the first byte, the second from those of the second byte: the implicit mod-32 of a shift by the
 * gram itself and by the gram >> 8 (scan_*.hip). */
inline uint32_t gram3Word(uint32_t key24, int log2Bits) { return ((uint32_t)((key24 & 0xFFFFFFu) * kGram3Mul) >> 18) & ((1u << (log2Bits - 5)) - 1u); }
inline uint32_t gram3Bit1(uint32_t key24) { return key24 & 31u; }
inline uint32_t gram3Bit2(uint32_t key24) { return (key24 >> 8) & 31u; }
/* gram1: the dword from bits 18..31 of the 24 x 24 -> 32 bit product (the same one-instruction address), the bit from the low
 * five bits of the gram's first byte (the implicit mod-32 of a shift by the gram itself) */
inline uint32_t gram1Word(uint32_t key24) { return ((uint32_t)((key24 & 0xFFFFFFu) * kGram1Mul) >> 18) & ((1u << (kGram1Log2 - 5)) - 1u); }
inline uint32_t gram1Bit(uint32_t key24) { return key24 & 31u; }
/* prefix4: keyed by the ladder's hash of the first four bytes, h = ladderStart(first4): bits h >> (32 - kPrefix4Log2) and
 * (h * kLadMulS) >> (32 - kPrefix4Log2) */
inline uint32_t prefix4Bit1(uint32_t h) { return h >> (32 - kPrefix4Log2); }
inline uint32_t prefix4Bit2(uint32_t h) { return (uint32_t)(h * kLadMulS) >> (32 - kPrefix4Log2); }
constexpr uint32_t kJumpMul = 0x9E3779B1u;    /* jump table of the chained walker (tables.cpp): slot of a 4-byte prefix */
constexpr int kJumpLog2Min = 10, kJumpLog2Max = 20;
inline uint32_t jumpHash(uint32_t key32, int log2Slots) { return (uint32_t)(key32 * kJumpMul) >> (32 - log2Slots); }
constexpr uint32_t kFinal3Mul = 0x85EBCBu, kFinal3Mul2 = 0xB5297Bu;    /* 24-bit odd multipliers of the two length-3 hashes */
inline uint32_t final3Hash(uint32_t key24, int log2Bits) { return (uint32_t)((key24 & 0xFFFFFFu) * kFinal3Mul) >> (32 - log2Bits); }
inline uint32_t final3Hash2(uint32_t key24, int log2Bits) { return (uint32_t)((key24 & 0xFFFFFFu) * kFinal3Mul2) >> (32 - log2Bits); }

} // namespace pfac

struct PFAC_context {
    /* compiled pattern set */
    pfac::Automaton fa;
    pfac::Filter filter;
    bool isPatternsReady = false;
    std::string patternFile;

    /* host tables (ref h_PFAC_table / h_hashRowPtr / h_hashValPtr / h_tableOfInitialState) */
    std::vector<int> h_dense;
    std::vector<pfac::Int2> h_hashRow;
    std::vector<pfac::Int2> h_hashVal;
    std::vector<int> h_initialRow;            /* 256 ints, valid in both modes */

    /* device tables */
    int *d_dense = nullptr;
    pfac::Int2 *d_hashRow = nullptr;
    pfac::Int2 *d_hashVal = nullptr;
    int *d_initialRow = nullptr;
    /* A small pattern set whose states hardly fold into chains (most states final or branching: the patterns a, aa, ..., a x 8 of the all-match test) gains
     * nothing from the chained table -- a step consumes one byte either way, and the chained step is three times the instructions of one gathered dword of
     * int[S][256].  Such a set also keeps the DENSE table on the device (S KiB, at most kDenseFastMaxStates states), in both perf modes, and
     * PFACX_KERNEL_AUTO sends the big calls of a pattern-dense stream to the tiled frame over it (scan_module.hip: scan) */
    int *d_denseFast = nullptr;
    size_t denseFastEntries = 0;
    std::vector<pfac::ChainSlot> h_chainSlots;               /* host copy of the chained table (PFACX_saveCompiled)       */
    pfac::ChainSlot *d_chainSlots = nullptr;  /* device-only chained form of hashRow/hashVal (tables.cpp)          */
    size_t numChainSlots = 0;
    /* the NARROW chained table (tables.cpp: no wide buckets, no long jump table, no units): what the tiled kernel walks while the handle's stream
     * is not full of near misses; device only (built from the trie with the set, dropped with it) */
    pfac::ChainSlot *d_chainNarrow = nullptr;
    size_t numChainNarrow = 0;
    int chainNarrowJumpLog2 = 0;
    int chainJumpLog2 = 0;                    /* the chained table is N = numChainSlots / 2 slot headers, then as many extension units; the last
                                                 2^J headers are the LONG jump table, the 2^J before them the jump table, the 256 before
                                                 those the initial state's bucket: buckets | root(256) | jump(2^J) | long jump(2^J) | N units
                                                 (tables.cpp: buildChainedHashTable) */
    uint32_t *d_gram3 = nullptr;
    uint32_t *d_shortBits = nullptr;
    uint32_t *d_ladder = nullptr;
    uint32_t *d_gram1 = nullptr, *d_prefix4 = nullptr;   /* the compacted-output kernel's level 1 and depth-4 test (struct Filter) */
    uint32_t *d_tail = nullptr;                          /* the tail table (struct Filter: `tail`, or `tailG` -- a set has one of them), or null */
    /* grow-only scratch of the compacted-output path (the arrays the pairs are ordered through), owned by the handle so that a
     * call does not pay for hipMalloc/hipFree */
    void *d_reduceScratch = nullptr;
    size_t reduceScratchBytes = 0;
    /* the counters at the head of the scratch as the last ordered call left them: all zero (the ordering launches clean up behind themselves, scan_order.inc), so
     * the next call with the same layout needs no memset in front of its scan; null = not known to be zero */
    const void *orderCleanBase = nullptr;
    size_t orderCleanBytes = 0;
    unsigned int orderSeq = 0;                /* number of the last ordered call (pfac_order_done writes it to host memory) */
    unsigned int orderParity = 0;             /* which of the two pairs of call counters the next ordered call uses */
    /* staging of PFAC_matchFromHost / PFAC_matchFromHostReduce on the GPU platform (pfac_api.cpp): two input and two result
     * buffers of hostStageChunk (+ overlap) positions, two copy streams, events; created on first use */
    char *d_stageIn[2] = {nullptr, nullptr};
    int *d_stageOut[2] = {nullptr, nullptr};
    bool reduceUnordered = false;             /* the compacted-output scan may leave its pairs in any order (set around the calls of PFAC_matchFromHost) */
    int *d_stagePos[2] = {nullptr, nullptr};  /* positions of the compacted results of a piece (PFAC_matchFromHost) */
    size_t hostStagePositions = 0;            /* capacity of each staging buffer, in positions */
    void *stageUp = nullptr, *stageDown = nullptr;                 /* hipStream_t */
    mutable bool countersDirty = false;                            /* a filter launch failed: the next one clears the launch counters itself */
    unsigned int denseParity = 0;                                  /* which of the two dense-chunk counters the next filter launch uses */
    /* PFACX_setKernelTiming: HIP events around the launch of the filter kernel (PFACX_getScanStats reports the time) */
    bool kernelTiming = false;
    void *evTime[2] = {nullptr, nullptr};                          /* hipEvent_t */
    mutable bool evTimeRecorded = false;
    void *evUp[2] = {nullptr, nullptr}, *evScan[2] = {nullptr, nullptr}, *evDown[2] = {nullptr, nullptr};   /* hipEvent_t */
    unsigned int *d_workCounters = nullptr;   /* kWorkCounterWords: next-chunk counters of the scan kernel (one per 128 B) */
    /* one word of mapped host memory the last block of a full-result filter launch writes: 1 = the stream was full of near misses
     * (scan_filter.hip: launchChained picks the next launch's walker from it); h_: the host's pointer, d_: the device's */
    unsigned int *h_modeHint = nullptr, *d_modeHint = nullptr;
    int walker = PFACX_WALKER_AUTO;
    uint32_t *d_final3 = nullptr;

    /* ref numOfTableEntry / sizeOfTableEntry / sizeOfTableInBytes, PFAC_P.h:131-133 */
    size_t numOfTableEntry = 0;
    size_t sizeOfTableEntry = 0;
    size_t sizeOfTableInBytes = 0;

    /* kernel module seam (ref PFAC_P.h:136-146) */
    void *module = nullptr;
    PFAC_kernel_protoType kernel_time_driven_ptr = nullptr;
    PFAC_kernel_protoType kernel_space_driven_ptr = nullptr;
    PFAC_reduce_kernel_protoType reduce_kernel_ptr = nullptr;
    PFAC_reduce_kernel_protoType reduce_inplace_kernel_ptr = nullptr;

    int platform = PFAC_PLATFORM_GPU;
    int perfMode = PFAC_TIME_DRIVEN;
    int textureMode = PFAC_AUTOMATIC;
    int kernelVariant = PFACX_KERNEL_AUTO;

    /* One handle may be shared by host threads (the reference serialises them with its texture mutex,
     * PFAC.cpp:37-56): every entry point that touches per-handle device state -- chunk counters, the counter and
     * scratch of the compacted-output path, the host staging buffers -- holds this lock for the call. */
    std::mutex lock;
    /* ... and the CPU platforms read the host tables for the whole of a match without it (several threads may match on the CPU
     * platforms at once): they hold this one shared; whoever frees or rebuilds the pattern set or its tables (readPattern*,
     * setPerfMode, loadCompiled, the dense table's first use) holds `lock` AND this one exclusively */
    std::shared_mutex tablesInUse;
    /* per-device handles of PFACX_matchFromHostMultiGPU, created on first use: (device, handle) */
    std::vector<std::pair<int, PFAC_context *>> children;
    /* chunks the filter kernel found pattern-dense and left to the simple kernel (scan_*.hip): grow-only, one entry per chunk of a launch */
    unsigned int *d_denseList = nullptr;
    size_t denseListEntries = 0;

    bool hasDevice = false;
    int device = -1;
    int multiProcessorCount = 0;
    std::string archName;
};

#endif /* PFAC_CONTEXT_H_ */

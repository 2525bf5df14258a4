/*
 * pfac_ext.h -- small extension surface next to the reference-compatible ABI
 * in PFAC.h.  Nothing here exists in the reference; every function is
 * prefixed PFACX_ so a drop-in user never sees it.  The extensions exist so
 * that tests and the bench harness can (a) exercise the host-side pattern
 * compiler on a machine without a GPU, (b) compare the compiled tables
 * byte-for-byte with the oracle, (c) read the facts a multi-GPU driver needs
 * (maximum pattern length = slice overlap, reference omp_PFAC.cpp:324) and
 * (d) A/B the kernel variants.
 */
#ifndef PFAC_EXT_H_
#define PFAC_EXT_H_

#include "PFAC.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Like PFAC_create() but binds no device and loads no kernel module: only
 * the host pattern compiler and the CPU platforms work.  Any GPU entry point
 * on such a handle returns PFAC_STATUS_LIB_NOT_EXIST -- there is no silent
 * CPU fallback for the GPU platform.  The platform is preset to
 * PFAC_PLATFORM_CPU. */
PFAC_status_t PFACX_createHostOnly(PFAC_handle_t *handle);

/* PFAC_readPatternFromFile (ref PFAC.cpp:653-735) for patterns that are already in memory: `patterns`
 * holds `size` bytes in the pattern-file format, one pattern per '\n'-terminated line (bytes after
 * the last '\n' are ignored, like in a file).  Same status codes, same pattern IDs; replaces a
 * previously loaded set.  The buffer is copied. */
PFAC_status_t PFACX_readPatternFromMemory(PFAC_handle_t handle, const char *patterns, size_t size);

/* The two readers with options (defined behaviour for what the reference does silently).
 *   PFACX_READ_STRICT    bytes behind the last '\n' -- a last line without a newline, which the reference drops without
 *                        a word (PFAC_reorder_Table.cpp:181-195) -- are PFAC_STATUS_INVALID_PARAMETER.  Without the flag
 *                        they are ignored as in the reference, and PFACX_getInfo reports how many there were
 *                        (trailingBytesIgnored).
 *   PFACX_READ_STRIP_CR  "\r\n" line ends: the '\r' is not part of the pattern (the reference keeps it: user guide r1.2
 *                        p.15 item 5).  A '\r' elsewhere in a line stays.
 * flags == 0 is exactly PFAC_readPatternFromFile / PFACX_readPatternFromMemory. */
#define PFACX_READ_STRICT   1u
#define PFACX_READ_STRIP_CR 2u
PFAC_status_t PFACX_readPatternFromFileEx(PFAC_handle_t handle, const char *filename, unsigned int flags);
PFAC_status_t PFACX_readPatternFromMemoryEx(PFAC_handle_t handle, const char *patterns, size_t size, unsigned int flags);

typedef struct {
    size_t structSize;        /* IN: sizeof(PFACX_info_t) of the caller's header; the library never writes past it (a
                                 caller built against an older, shorter struct stays safe).  OUT: bytes filled in */
    int numOfPatterns;        /* F                                             */
    int numOfStates;          /* includes the unused state 0 (ref PFAC.cpp:704) */
    int numOfFinalStates;     /* == numOfPatterns                               */
    int initialState;         /* F + 1                                          */
    int maxPatternLen;
    int numOfLeaves;
    int perfMode;             /* PFAC_perfMode_t                                */
    int textureMode;          /* PFAC_textureMode_t as currently stored         */
    int platform;             /* PFAC_platform_t                                */
    int hasDevice;            /* 0 for PFACX_createHostOnly handles             */
    size_t numOfTableEntry;   /* ref PFAC_P.h:131-133                           */
    size_t sizeOfTableEntry;
    size_t sizeOfTableInBytes;
    /* prefilter (this implementation only; see DESIGN.md "filter") */
    int filterLog2Bits;       /* 3-gram bitmap has 2^filterLog2Bits bits (two per 3-gram, in one dword) */
    int filterHasShort;       /* 1 if some pattern is shorter than 3 bytes      */
    size_t filterBitsSet;     /* population of the 3-gram bitmap                */
    int kernelVariant;        /* PFACX_KERNEL_*                                 */
    int multiProcessorCount;
    int filterLog2BitsLadder; /* prefix-ladder bitmap has 2^filterLog2BitsLadder bits (level 2 of the prefilter) */
    int filterLog2BitsFinal3; /* length-3-pattern bitmap                         */
    size_t filterBitsSetLadder;
    int chainJumpLog2;        /* PFACX_TABLE_CHAIN: log2 of its jump-table slots (0 until the table exists) */
    size_t chainSlots;        /* PFACX_TABLE_CHAIN: 16-byte units in total: slot headers, then as many extension units */
    size_t ladderStops;       /* prefix ladder: trie nodes inserted as "stop: walk from here" ...            */
    size_t ladderGoOns;       /* ... and as "go on: test the next prefix length"                             */
    int ladderThin;           /* nodes with at most this many patterns below them are stops ...             */
    int ladderExtend;         /* ... this many levels further down                                          */
    size_t trailingBytesIgnored; /* bytes behind the last '\n' of the pattern file that were ignored (0: none)  */
    size_t deviceTableBytes;  /* device memory the pattern set holds: chained table, initial row, prefilter bitmaps,
                                 launch counters; the reference-layout table only while PFACX_KERNEL_REFTABLE is selected */
    size_t deviceScratchBytes; /* device memory the handle's calls have left allocated (grow-only; PFACX_trim frees it): the two staging
                                 pieces of PFAC_matchFromHost / ...Reduce (9 bytes per position of a piece), the ordering scratch of the
                                 compacted output, the list of pattern-dense chunks */
    int streamNearMisses;      /* what the handle's last big full-result launch said about its stream (host memory the launch's last block
                                 writes; nothing is waited for): 1 = full of near misses of long patterns -> PFACX_WALKER_AUTO picks STAGE ... */
    int streamDense;           /* ... 1 = most of it pattern-dense (short patterns over text, runs of a pattern byte) -> PFACX_KERNEL_AUTO
                                 sends the next big call to the tiled kernel alone */
    int filterLadderLast;      /* deepest level of the prefix ladder: 20, or 60 when the nodes behind the 20th byte fit its bitmap too */
    size_t filterTailEntries;  /* entries of the tail table in its LDS form (PFACX_TABLE_FILTER_TAIL): sets of a few thousand patterns */
    size_t filterTailGlobalEntries; /* ... in its device-memory form (PFACX_TABLE_FILTER_TAIL_GLOBAL): sets whose bitmaps fill the LDS, or with more
                                  thin stops than the LDS form holds (Snort-scale).  A set has one form or the other (or none) */
    int filterLog2TailGlobal;  /* log2 of the buckets of that table */
    unsigned int filterLadderSalt; /* XORed into the depth-4 hash of the prefix ladder (chosen per pattern set: pfac_context.h) */
    int filterSkipTags;        /* skip tags of the prefix ladder (PFACX_TABLE_FILTER_SKIP): depth-6 nodes with a single path down to depth 20 (at most 8) */
} PFACX_info_t;

PFAC_status_t PFACX_getInfo(PFAC_handle_t handle, PFACX_info_t *info);

/* Host copies of the compiled tables, owned by the handle (valid until the
 * next readPatternFromFile / setPerfMode / destroy). */
typedef enum {
    PFACX_TABLE_DENSE        = 0,  /* int[numOfStates*256]           (TIME_DRIVEN)  */
    PFACX_TABLE_HASH_ROWPTR  = 1,  /* int2[numOfStates]              (SPACE_DRIVEN) */
    PFACX_TABLE_HASH_VALPTR  = 2,  /* int2[numOfTableEntry]          (SPACE_DRIVEN) */
    PFACX_TABLE_INITIAL_ROW  = 3,  /* int[256], both modes                          */
    PFACX_TABLE_FILTER_GRAM3 = 4,  /* uint32[2^filterLog2Bits / 32]                 */
    PFACX_TABLE_FILTER_SHORT = 5,  /* uint32[2048] (65536 bits)                     */
    PFACX_TABLE_FILTER_LADDER = 6, /* uint32[2^filterLog2BitsLadder / 32]           */
    PFACX_TABLE_FILTER_FINAL3 = 7, /* uint32[2^filterLog2BitsFinal3 / 32]           */
    PFACX_TABLE_FILTER_GRAM1 = 9,  /* uint32[2^19 / 32]: one-bit 3-gram bitmap of the compacted-output kernel       */
    PFACX_TABLE_FILTER_PREFIX4 = 10, /* uint32[2^17 / 32]: the 4-byte pattern prefixes, two probes (same kernel)    */
    PFACX_TABLE_FILTER_TAIL  = 11, /* uint32[3] per slot {ladder hash of a stop node, that hash rolled over the rest of the one pattern below it,
                                      bytes of that rest | depth << 8}, a power of two of slots (none: empty): the veto on a ladder stop */
    PFACX_TABLE_FILTER_TAIL_GLOBAL = 12, /* uint32[4] per bucket: two entries {ladder hash of a stop node, (that hash rolled over the rest of the one
                                      pattern below it) & ~0x7FF | depth of the first compared byte << 3 | bytes / 4 - 1}; bucket of a hash h =
                                      (h * 0x9E3779B1) >> (32 - filterLog2TailGlobal); an entry is occupied if (word1 & 0x7F8) != 0 */
    PFACX_TABLE_FILTER_SKIP  = 13, /* uint32[filterSkipTags]: ladder hashes (depth 6) whose candidates are next asked at depth 20 */
    PFACX_TABLE_CHAIN        = 8   /* uint32[4] per 16-byte unit: the device-only chained form of the hashed table that the
                                      GPU kernels walk in both perf modes.  chainSlots / 2 slot headers -- compact buckets,
                                      breadth first; then the 256 slots of the initial state; then the 2^chainJumpLog2
                                      slots of the jump table of 4-byte prefixes; then the LONG jump table, 2^chainJumpLog2
                                      slots again (same hash, chains of up to 23 bytes) -- followed by as many extension
                                      units, unit i = chain bytes 8..22 of slot i (long slots of wide buckets and of the long
                                      jump table only).  With N = chainSlots / 2 and J = chainJumpLog2:
                                        [0, N - 256 - 2 * 2^J) buckets | [.., N - 2 * 2^J) initial state | [.., N - 2^J) jump |
                                        [.., N) long jump | [N, 2 N) extension units.
                                      Built on first use on a host-only handle.  */
} PFACX_table_t;

PFAC_status_t PFACX_getTable(PFAC_handle_t handle, PFACX_table_t which, const void **ptr,
                             size_t *bytes);

/* Kernel variants of the GPU match path. */
/* The walker of the full-result filter kernel (PFACX_setWalker).  WINDOW: a walk's input travels with its queue entry and
 * lives in registers (fastest on text); STAGE: walks read their input in place from the chunks a wave keeps staged in LDS and
 * take 24 bytes per step through long single-successor runs (fastest when the stream is full of near misses of long
 * patterns: BASELINE config 5, 19 % over WINDOW); AUTO (default): whatever the handle's previous full-result launch found its
 * stream to be -- the first launch on a handle runs WINDOW.  Results are identical.
 * A pattern set that has a TAIL TABLE (filterTailEntries of PFACX_getInfo > 0 and room in the CU's LDS: sets of a few thousand
 * patterns) is different: the prefilter puts a ladder stop to that table before it becomes a walk, near misses hardly reach a walker,
 * and AUTO and WINDOW both mean the window walker behind that veto (another 13 % on BASELINE config 5); STAGE is the stage walker
 * without it.
 * A set whose tail table lies in DEVICE memory (filterTailGlobalEntries > 0: Snort-scale sets, round 6) asks it with one gathered load per
 * stopped candidate, which text does not repay: under AUTO such a set runs the plain window walker until a launch reports a stream full of
 * near misses, and the veto kernel (not the stage walker) from then on, until a launch of that kernel reports text again; WINDOW is the
 * plain window walker, VETO the veto kernel, always. */
#define PFACX_WALKER_AUTO   0
#define PFACX_WALKER_WINDOW 1
#define PFACX_WALKER_STAGE  2
#define PFACX_WALKER_VETO   3   /* the window walker behind the tail-hash veto whatever the stream looks like (a set without a tail table of either form: WINDOW) */
PFAC_status_t PFACX_setWalker(PFAC_handle_t handle, int walker);

#define PFACX_KERNEL_FILTER 0   /* LDS prefilter + compacted walkers wherever the pointers allow it */
#define PFACX_KERNEL_NAIVE  1   /* the tiled kernel alone: one position per thread slot, tile + halo and the
                                   hottest transition rows in LDS, coalesced result lines (any alignment)   */
#define PFACX_KERNEL_AUTO   2   /* default: FILTER, except that small calls take the tiled kernel alone
                                   (lower latency: the filter kernel has a ~19 us floor).  Either way the
                                   filter kernel hands pattern-dense 2 KiB chunks (most positions pass its
                                   first level) to the tiled kernel that follows it.                      */
#define PFACX_KERNEL_REFTABLE 3 /* the reference-shaped kernel: one thread per byte through the REFERENCE-layout
                                   table of the perf mode (dense int[S][256] / hashed int2 pair), which is built
                                   and uploaded when this variant is selected (S KiB for the dense table).  An
                                   independent second implementation for cross-checks; 6-20x slower.        */

PFAC_status_t PFACX_setKernelVariant(PFAC_handle_t handle, int variant);

/* A compiled pattern set on disk: everything PFAC_readPatternFromFile derives from the pattern
 * file -- trie, hashed / chained tables, prefilter bitmaps -- with a version + layout fingerprint + checksum
 * header.  PFACX_loadCompiled replaces the handle's pattern set like PFAC_readPatternFromFile does (and sets the
 * perf mode the set was saved with); a file of another build or a damaged one is PFAC_STATUS_INVALID_PARAMETER,
 * a missing one PFAC_STATUS_FILE_OPEN_ERROR.  Works on host-only handles too. */
PFAC_status_t PFACX_saveCompiled(PFAC_handle_t handle, const char *filename);
PFAC_status_t PFACX_loadCompiled(PFAC_handle_t handle, const char *filename);

/* The handle keeps its device temporaries between calls (the reference allocates and frees them per call,
 * PFAC.cpp:920-958): two staging sets of PFAC_matchFromHost (about 9 B per position of a 32 Mi-position piece), the copies
 * of PFAC_matchFromHostReduce (9 B per input byte), sort scratch.  PFACX_trim frees them; the next call that needs one
 * allocates it again.  The pattern set and its tables stay. */
PFAC_status_t PFACX_trim(PFAC_handle_t handle);

/* What the handle's FIRST PFAC_matchFromHost / PFAC_matchFromHostReduce would allocate, create and load inside the call, done ahead of it: the
 * two staging pieces (for calls of up to maxBytes input bytes; 0 or more than a piece: whole pieces of 32 Mi positions), their copy streams and
 * events, the ordering scratch, the kernels' code objects (one throwaway scan), the runtime's staging of pageable host memory (one throwaway
 * upload).  Optional: without it the first call does the same, ~100 ms instead of ~7 (256 MiB).  PFACX_trim undoes it; a handle on a CPU
 * platform has nothing to prepare (success). */
PFAC_status_t PFACX_prepare(PFAC_handle_t handle, size_t maxBytes);

/* One call shards a host stream over several GPUs of the node (reference users write this
 * themselves after PFAC/test/omp_PFAC.cpp:257-394): one worker thread and one internal handle per entry of
 * `devices` (NULL = devices 0..numDevices-1; numDevices 0 = every visible device), contiguous slices scanned
 * with a maxPatternLen read-ahead, no exchange between devices.  `handle` supplies the pattern set and modes
 * and keeps the per-device handles for the next call.  A device may be listed more than once.
 * On a handle whose platform is a CPU platform (PFAC_setPlatform, PFACX_createHostOnly) there is nothing to shard over: the call then runs
 * `numDevices` workers (0 = one) as host threads over the CPU matchers -- same slices, same read-ahead, same folding of the results; `devices`
 * is ignored.  That is a dry run of the driver on a machine without a GPU, not a fallback: the GPU platform never takes it. */
PFAC_status_t PFACX_matchFromHostMultiGPU(PFAC_handle_t handle, char *h_inputString, size_t size,
                                          int *h_matched_result, int numDevices, const int *devices);

/* ... and its compacted-output form (PFAC_matchFromHostReduce over several GPUs): the (id, position) pairs of the whole stream in ascending
 * position order, positions counted from the start of the stream; h_matched_result and h_pos hold `size` entries each (the workers use
 * them as scratch), size < 2^31.  One byte per position crosses a host link and nothing is filled on the host: this, not the
 * full-vector form, is what scales with the number of links. */
PFAC_status_t PFACX_matchFromHostReduceMultiGPU(PFAC_handle_t handle, char *h_inputString, size_t size, int *h_matched_result, int *h_pos,
                                                int *h_num_matched, int numDevices, const int *devices);

/* Counters of the most recent launch of the filter kernel on this handle (PFAC_matchFromDevice / ...Reduce of
 * 32 MiB or more; SURVEY 8d, configuration C5: walk depth, lane utilisation, early-out rate).  Waits for the default
 * stream.  All zero before the first such launch. */
typedef struct {
    size_t structSize;                    /* IN: sizeof(PFACX_scan_stats_t) of the caller's header; OUT: bytes filled in (see PFACX_info_t) */
    unsigned long long walkerRounds;      /* wave-wide walker rounds (one table step for every live walk)      */
    unsigned long long laneSteps;         /* table steps taken, summed over lanes                              */
    unsigned long long walksStarted;      /* positions that passed level 1 and the prefix ladder and were walked */
    unsigned long long level1Hits;        /* positions that passed filter level 1                              */
    int tilesPerChunk;                    /* KiB per chunk                                                     */
    int walksPerLane;                     /* independent walks per lane in THAT launch (the full-result and the compacted-
                                             output kernel differ)                                              */
    unsigned long long ladderCandidates;  /* level-1 hits whose first four bytes are a pattern prefix (or a short
                                             pattern): what the prefix ladder was asked about                  */
    unsigned long long denseChunks;       /* 2 KiB chunks in which more than half of the positions (PFAC_DENSE_HITS of 2048) passed level 1:
                                             left to the tiled kernel that follows the filter kernel (0 for a
                                             compacted-output launch, which lists none)                         */
    double filterKernelMs;                /* GPU time of that launch of the filter kernel alone (HIP events around it);
                                             -1 unless PFACX_setKernelTiming(handle, 1) was in force           */
    unsigned long long stageModeWaves;    /* full-result launch: scanning waves that ended it finding their stream full of near misses
                                             (STAGE walker: in stage mode, two chunks staged, walking out of LDS; WINDOW walker:
                                             fetching extension units ahead); a majority makes PFACX_WALKER_AUTO give the handle's
                                             next launch the STAGE walker                                          */
    int walker;                           /* PFACX_WALKER_WINDOW / PFACX_WALKER_STAGE: what that launch ran with */
    int veto;                             /* ... and whether the window walker stood behind the tail-hash veto: 0 = no, 1 = tail table in LDS, 2 = in device
                                             memory (the kernel PFACX_WALKER_VETO asks for) */
} PFACX_scan_stats_t;

PFAC_status_t PFACX_getScanStats(PFAC_handle_t handle, PFACX_scan_stats_t *stats);

/* Measurement aid: with `on` != 0 every launch of the filter kernel on this handle is bracketed by two HIP events on
 * the default stream (a few microseconds per call); PFACX_getScanStats then reports the kernel's own time -- inside
 * PFAC_matchFromDeviceReduce, say, whose other launches a caller's events cannot tell apart. */
PFAC_status_t PFACX_setKernelTiming(PFAC_handle_t handle, int on);

#ifdef __cplusplus
}
#endif

#endif /* PFAC_EXT_H_ */

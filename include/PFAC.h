/*
 * PFAC.h -- public C ABI of the MI355X-native PFAC library (libpfac.so).
 *
 * PFAC = Parallel Failureless Aho-Corasick: exact multi-pattern string
 * matching where position j of the input reports the ID (1-based, order of
 * appearance in the pattern file) of the LONGEST pattern that starts at j,
 * or 0 when no pattern starts there.
 *
 * This header is the drop-in boundary: every name, signature, enumerator and
 * enumerator VALUE below is binary compatible with the reference
 * (pfac-lib/PFAC r1.2, PFAC/include/PFAC.h:27-215), so a program compiled
 * against the reference header links against this library unchanged.  The
 * body of the library is a from-scratch HIP/gfx950 implementation; see
 * DESIGN.md.  Each declaration cites the reference interface it replaces.
 *
 * Memory spaces: "device" pointers are hipMalloc'd (or any device-accessible)
 * addresses on the GPU that was current when PFAC_create() ran.  The library
 * never calls hipSetDevice().
 */
#ifndef PFAC_H_
#define PFAC_H_

#include <stdio.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Where PFAC_matchFromHost() runs (ref PFAC.h:27-31).  matchFromDevice is
 * always a GPU call. */
typedef enum {
    PFAC_PLATFORM_GPU     = 0,   /* default */
    PFAC_PLATFORM_CPU     = 1,   /* one host thread */
    PFAC_PLATFORM_CPU_OMP = 2    /* OpenMP; needs OMP_NUM_THREADS in the environment,
                                    otherwise behaves like PFAC_PLATFORM_CPU */
} PFAC_platform_t;

/* How the GPU kernel reads the transition table (ref PFAC.h:33-37).
 * gfx950 has no texture path for linear memory; TEXTURE_ON selects the
 * closest CDNA4 analogue, a bounds-checked read-only buffer-resource
 * descriptor (buffer_load), TEXTURE_OFF selects plain global loads.
 * AUTOMATIC resolves to ON when the table has fewer than 2^27 entries. */
typedef enum {
    PFAC_AUTOMATIC   = 0,        /* default */
    PFAC_TEXTURE_ON  = 1,
    PFAC_TEXTURE_OFF = 2
} PFAC_textureMode_t;

/* Transition-table representation (ref PFAC.h:39-42). */
typedef enum {
    PFAC_TIME_DRIVEN  = 0,       /* default: dense numOfStates x 256 int table        */
    PFAC_SPACE_DRIVEN = 1        /* per-state perfect hash (k*c mod 257 mod S), ~1/50 */
} PFAC_perfMode_t;

/* Status codes (ref PFAC.h:57-70).  Values below PFAC_STATUS_BASE other than
 * 0 are raw hipError_t values forwarded from the HIP runtime (the reference
 * forwards cudaError_t the same way, PFAC.cpp:148-151). */
typedef enum {
    PFAC_STATUS_SUCCESS            = 0,
    PFAC_STATUS_BASE               = 10000,
    PFAC_STATUS_ALLOC_FAILED       = 10001,  /* host allocation failed                 */
    PFAC_STATUS_CUDA_ALLOC_FAILED  = 10002,  /* device allocation failed (name kept)   */
    PFAC_STATUS_INVALID_HANDLE     = 10003,
    PFAC_STATUS_INVALID_PARAMETER  = 10004,
    PFAC_STATUS_PATTERNS_NOT_READY = 10005,
    PFAC_STATUS_FILE_OPEN_ERROR    = 10006,
    PFAC_STATUS_LIB_NOT_EXIST      = 10007,  /* kernel module libpfac_<arch>.so missing */
    PFAC_STATUS_ARCH_MISMATCH      = 10008,
    PFAC_STATUS_MUTEX_ERROR        = 10009,
    PFAC_STATUS_INTERNAL_ERROR     = 10010
} PFAC_status_t;

struct PFAC_context;
typedef struct PFAC_context *PFAC_handle_t;

/* ref PFAC.h:87, PFAC.cpp:133-204.  Allocates a context bound to the current
 * HIP device and loads the kernel module for its architecture
 * (libpfac_gfx950.so).  Returns a raw hipError_t if no device is usable,
 * PFAC_STATUS_LIB_NOT_EXIST if the module cannot be loaded. */
PFAC_status_t PFAC_create(PFAC_handle_t *handle);

/* ref PFAC.h:96, PFAC.cpp:207-218.  Frees host and device tables and the context. */
PFAC_status_t PFAC_destroy(PFAC_handle_t handle);

/* ref PFAC.h:106, PFAC.cpp:741-757. */
PFAC_status_t PFAC_setPlatform(PFAC_handle_t handle, PFAC_platform_t platform);

/* ref PFAC.h:120, PFAC.cpp:764-779. */
PFAC_status_t PFAC_setTextureMode(PFAC_handle_t handle, PFAC_textureMode_t textureModeSel);

/* ref PFAC.h:131, PFAC.cpp:782-817.  Changing the mode after patterns are
 * loaded rebuilds and re-uploads the transition table. */
PFAC_status_t PFAC_setPerfMode(PFAC_handle_t handle, PFAC_perfMode_t perfModeSel);

/* ref PFAC.h:140, PFAC.cpp:1131-1183.  Static storage; do not free or modify. */
const char *PFAC_getErrorString(PFAC_status_t status);

/* ref PFAC.h:149, PFAC.cpp:1188-1246.  Text dump: "(state, char) -> next"
 * lines then the final-state/pattern table; fp == NULL means stdout. */
PFAC_status_t PFAC_dumpTransitionTable(PFAC_handle_t handle, FILE *fp);

/* ref PFAC.h:166, PFAC.cpp:653-735.  One pattern per '\n'-terminated line,
 * arbitrary bytes except '\n'; IDs are 1,2,... in file order.  Loading again
 * replaces the previous pattern set. */
PFAC_status_t PFAC_readPatternFromFile(PFAC_handle_t handle, char *filename);

/* ref PFAC.h:179, PFAC.cpp:843-876.  d_inputString: size bytes on the device,
 * d_matched_result: size ints on the device, every element is written.
 * Asynchronous on the default stream.  size == 0 is a successful no-op. */
PFAC_status_t PFAC_matchFromDevice(PFAC_handle_t handle, char *d_inputString, size_t size,
                                   int *d_matched_result);

/* ref PFAC.h:198, PFAC.cpp:879-961.  Host buffers; runs on the platform
 * selected by PFAC_setPlatform().  Synchronous. */
PFAC_status_t PFAC_matchFromHost(PFAC_handle_t handle, char *h_inputString, size_t size,
                                 int *h_matched_result);

/* ref PFAC.h:206, PFAC.cpp:964-1008.  Compacted output: the first
 * *h_num_matched entries of d_matched_result / d_pos hold the non-zero
 * results and their positions in ascending position order. */
PFAC_status_t PFAC_matchFromDeviceReduce(PFAC_handle_t handle, char *d_inputString, size_t size,
                                         int *d_matched_result, int *d_pos, int *h_num_matched);

/* ref PFAC.h:214, PFAC.cpp:1010-1128. */
PFAC_status_t PFAC_matchFromHostReduce(PFAC_handle_t handle, char *h_inputString, size_t size,
                                       int *h_matched_result, int *h_pos, int *h_num_matched);

#ifdef __cplusplus
}
#endif

#endif /* PFAC_H_ */

/*
 * pfac_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference algorithm for the PFAC match
 * path: pattern-file parser, trie builder with the reference's state
 * numbering, dense and perfect-hash table materialisers, scalar and OpenMP
 * matchers, and the transition-table dump.  It is the checker the parity
 * tests compare the HIP path against; it is never linked into, loaded by or
 * called from the product library (pfac_amd/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Pinning: the restatement is checked (tests/test_oracle_golden.py) against
 * the reference's own known answers -- README.md:113-120, user guide r1.2
 * p.21 (table dump, byte for byte), p.27, p.29, PFAC_hash_draft.pdf Fig.1 --
 * and against the reference's real parser / trie builder / CPU matchers
 * compiled unmodified into oracle/_ref (see oracle/Makefile).
 *
 * Status codes returned are the numeric PFAC_status_t values
 * (/root/reference/PFAC/include/PFAC.h:57-70).
 */
#ifndef PFAC_ORACLE_H_
#define PFAC_ORACLE_H_

#include <stdio.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_TRAP_STATE (-1)          /* 0xFFFFFFFF, ref PFAC_P.h:182 */
#define ORACLE_CHAR_SET   256           /* ref PFAC_P.h:181 */

enum {
    ORACLE_SUCCESS = 0,
    ORACLE_ALLOC_FAILED = 10001,
    ORACLE_INVALID_PARAMETER = 10004,
    ORACLE_PATTERNS_NOT_READY = 10005,
    ORACLE_FILE_OPEN_ERROR = 10006,
    ORACLE_INTERNAL_ERROR = 10010
};

typedef struct { int next_state; int ch; } oracle_edge_t;      /* ref TableEle, PFAC_P.h:51-54 */
typedef struct { int x; int y; } oracle_int2_t;                /* layout of CUDA int2 */

typedef struct oracle_pfac {
    /* pattern set (ref parsePatternFile outputs, PFAC_reorder_Table.cpp:121-231) */
    unsigned char *file;         /* raw bytes of the pattern file                      */
    long file_size;
    int num_patterns;            /* F                                                  */
    int *sorted_off;             /* [F] file offset of i-th pattern in sorted order    */
    int *sorted_id;              /* [F] pattern ID (1-based file order) of the same    */
    int *pattern_len;            /* [F+1] by ID, [0] = 0                               */
    int *pattern_off;            /* [F+1] by ID                                        */
    int max_pattern_len;
    /* automaton (ref create_PFACTable_spaceDriven, PFAC_reorder_Table.cpp:256-329) */
    int initial_state;           /* F+1                                                */
    int num_states;              /* next unused ID; counts the unused state 0          */
    int num_leaves;
    oracle_edge_t **row;         /* [num_states] edges in insertion order              */
    int *row_n;
    int *row_cap;
    int rows_alloc;
    /* dense table (ref PFAC_create2DTable, PFAC.cpp:345-402) */
    int *dense;                  /* [num_states*256]                                   */
    /* hashed table (ref PFAC_createHashTable, PFAC.cpp:422-648) */
    oracle_int2_t *hash_row;     /* [num_states] {offset, (k<<16)|(S-1)} or {-1,-1}    */
    oracle_int2_t *hash_val;     /* [hash_total] {next, ch}, unused slots {-1,-1}      */
    long hash_total;
    int *initial_row;            /* [256]                                              */
} oracle_pfac_t;

/* parse + sort + build the trie.  *out is NULL on failure. */
int oracle_load(const char *pattern_file, oracle_pfac_t **out);
void oracle_free(oracle_pfac_t *o);

int oracle_build_dense(oracle_pfac_t *o);
int oracle_build_hash(oracle_pfac_t *o);

/* ref PFAC_CPU_timeDriven / PFAC_CPU_spaceDriven (PFAC_CPU.cpp:60-163) */
int oracle_match_dense(const oracle_pfac_t *o, const unsigned char *in, size_t n, int *out);
int oracle_match_hash(const oracle_pfac_t *o, const unsigned char *in, size_t n, int *out);
/* ref PFAC_CPU_OMP_timeDriven / _spaceDriven (PFAC_CPU_OMP.cpp:81-185); nthreads<=0: runtime default */
int oracle_match_dense_omp(const oracle_pfac_t *o, const unsigned char *in, size_t n, int *out,
                           int nthreads);
int oracle_match_hash_omp(const oracle_pfac_t *o, const unsigned char *in, size_t n, int *out,
                          int nthreads);
int oracle_omp_max_threads(void);

/* ref PFAC_dumpTransitionTable (PFAC.cpp:1188-1246) */
int oracle_dump_table(const oracle_pfac_t *o, FILE *fp);
int oracle_dump_table_to_path(const oracle_pfac_t *o, const char *path);

/* ref the zip loop of PFAC_matchFromHostReduce CPU branch (PFAC.cpp:1055-1066):
 * compacts result in place, writes positions, returns count. */
long oracle_reduce(int *result, int *pos, size_t n);

/* 64-bit FNV-1a over the little-endian int32 result array and the number of
 * non-zero entries: the digest used for GiB-scale parity. */
void oracle_digest(const int *result, size_t n, unsigned long long *fnv, unsigned long long *count);

#ifdef __cplusplus
}
#endif
#endif

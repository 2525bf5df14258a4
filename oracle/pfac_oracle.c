/*
 * pfac_oracle.c -- TEST INFRASTRUCTURE ONLY (see pfac_oracle.h).
 *
 * Plain-C restatement of the reference's CPU path.  Every function cites the
 * reference lines it follows (paths relative to /root/reference/).  The code
 * is written for obviousness, not speed; the OpenMP matchers keep the
 * reference's structure (serial zero fill, then a static-schedule parallel
 * for over start positions) because they are also the timed "port" baseline.
 */
#include "pfac_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ parse */

static const unsigned char *g_sort_base; /* qsort has no context argument */

/* ref pattern_cmp_functor, PFAC/src/PFAC_reorder_Table.cpp:37-72:
 * bytes compared as SIGNED char, a proper prefix sorts first.  Patterns are
 * '\n'-terminated inside the file buffer.  Equal patterns compare equal here
 * (the reference returns true both ways, which is undefined for std::sort;
 * oracle_load rejects duplicates instead). */
static int cmp_patterns(const void *pa, const void *pb)
{
    const unsigned char *s = g_sort_base + *(const int *)pa;
    const unsigned char *t = g_sort_base + *(const int *)pb;
    for (;;) {
        signed char sc = (signed char)*s++;
        signed char tc = (signed char)*t++;
        int s_end = (sc == '\n');
        int t_end = (tc == '\n');
        if (s_end || t_end) {
            if (s_end == t_end) return 0;
            return s_end ? -1 : 1;
        }
        if (sc < tc) return -1;
        if (sc > tc) return 1;
    }
}

static int row_push(oracle_pfac_t *o, int state, int ch, int next)
{
    if (state >= o->rows_alloc) {
        int want = o->rows_alloc ? o->rows_alloc : 64;
        while (want <= state) want *= 2;
        oracle_edge_t **r = (oracle_edge_t **)realloc(o->row, sizeof(*r) * want);
        int *n = (int *)realloc(o->row_n, sizeof(int) * want);
        int *c = (int *)realloc(o->row_cap, sizeof(int) * want);
        if (!r || !n || !c) return ORACLE_ALLOC_FAILED;
        for (int i = o->rows_alloc; i < want; i++) { r[i] = NULL; n[i] = 0; c[i] = 0; }
        o->row = r; o->row_n = n; o->row_cap = c; o->rows_alloc = want;
    }
    if (o->row_n[state] == o->row_cap[state]) {
        int cap = o->row_cap[state] ? o->row_cap[state] * 2 : 2;
        oracle_edge_t *e = (oracle_edge_t *)realloc(o->row[state], sizeof(*e) * cap);
        if (!e) return ORACLE_ALLOC_FAILED;
        o->row[state] = e; o->row_cap[state] = cap;
    }
    o->row[state][o->row_n[state]].ch = ch;
    o->row[state][o->row_n[state]].next_state = next;
    o->row_n[state]++;
    return ORACLE_SUCCESS;
}

/* ref lookup(), PFAC_reorder_Table.cpp:234-244: first edge with this byte. */
static int row_lookup(const oracle_pfac_t *o, int state, int ch)
{
    if (state >= o->rows_alloc) return ORACLE_TRAP_STATE;
    for (int j = 0; j < o->row_n[state]; j++)
        if (o->row[state][j].ch == ch) return o->row[state][j].next_state;
    return ORACLE_TRAP_STATE;
}

void oracle_free(oracle_pfac_t *o)
{
    if (!o) return;
    for (int i = 0; i < o->rows_alloc; i++) free(o->row[i]);
    free(o->row); free(o->row_n); free(o->row_cap);
    free(o->file); free(o->sorted_off); free(o->sorted_id);
    free(o->pattern_len); free(o->pattern_off);
    free(o->dense); free(o->hash_row); free(o->hash_val); free(o->initial_row);
    free(o);
}

int oracle_load(const char *pattern_file, oracle_pfac_t **out)
{
    if (!out) return ORACLE_INVALID_PARAMETER;
    *out = NULL;
    if (!pattern_file) return ORACLE_INVALID_PARAMETER;           /* ref :125-127 */
    FILE *fp = fopen(pattern_file, "rb");
    if (!fp) return ORACLE_FILE_OPEN_ERROR;                       /* ref :146-150 */

    oracle_pfac_t *o = (oracle_pfac_t *)calloc(1, sizeof(*o));
    if (!o) { fclose(fp); return ORACLE_ALLOC_FAILED; }
    fseek(fp, 0, SEEK_END);
    long fsz = ftell(fp);
    rewind(fp);
    o->file = (unsigned char *)malloc(fsz > 0 ? (size_t)fsz : 1);
    if (!o->file) { fclose(fp); oracle_free(o); return ORACLE_ALLOC_FAILED; }
    fsz = (long)fread(o->file, 1, (size_t)fsz, fp);               /* ref :165 */
    fclose(fp);
    o->file_size = fsz;

    /* ref :168-193: a pattern ends at a '\n' whose predecessor is not '\n';
     * IDs count up in file order; bytes after the last '\n' are ignored; the
     * start pointer only advances at a non-empty line, so a blank line makes
     * the NEXT pattern begin with '\n' and the reference asserts (:291).
     * The oracle reports that case as ORACLE_INVALID_PARAMETER.  Blank lines
     * after the last pattern are harmless in the reference and here. */
    int cap = 16, count = 0;
    int *off = (int *)malloc(sizeof(int) * cap);
    int *len = (int *)malloc(sizeof(int) * cap);
    if (!off || !len) { free(off); free(len); oracle_free(o); return ORACLE_ALLOC_FAILED; }
    int start = 0, cur = 0, blank = 0;
    for (long i = 0; i < fsz; i++) {
        if (o->file[i] == '\n') {
            if (i > 0 && o->file[i - 1] != '\n') {
                if (o->file[start] == '\n') blank = 1;   /* stale start after a blank line */
                if (count == cap) {
                    cap *= 2;
                    off = (int *)realloc(off, sizeof(int) * cap);
                    len = (int *)realloc(len, sizeof(int) * cap);
                    if (!off || !len) { free(off); free(len); oracle_free(o); return ORACLE_ALLOC_FAILED; }
                }
                off[count] = start; len[count] = cur; count++;
                start = (int)i + 1;
            }
            cur = 0;
        } else {
            cur++;
        }
    }
    if (blank) { free(off); free(len); oracle_free(o); return ORACLE_INVALID_PARAMETER; }

    o->num_patterns = count;                                      /* ref :195 */
    o->sorted_off = (int *)malloc(sizeof(int) * (count ? count : 1));
    o->sorted_id = (int *)malloc(sizeof(int) * (count ? count : 1));
    o->pattern_len = (int *)calloc((size_t)count + 1, sizeof(int));
    o->pattern_off = (int *)calloc((size_t)count + 1, sizeof(int));
    if (!o->sorted_off || !o->sorted_id || !o->pattern_len || !o->pattern_off) {
        free(off); free(len); oracle_free(o); return ORACLE_ALLOC_FAILED;
    }
    o->max_pattern_len = 0;                                       /* ref PFAC.cpp:686-691 */
    for (int i = 0; i < count; i++) {
        o->pattern_len[i + 1] = len[i];
        o->pattern_off[i + 1] = off[i];
        if (len[i] > o->max_pattern_len) o->max_pattern_len = len[i];
        o->sorted_off[i] = off[i];
    }
    free(len);
    g_sort_base = o->file;
    qsort(o->sorted_off, (size_t)count, sizeof(int), cmp_patterns);   /* ref :200 */
    /* map sorted offsets back to IDs (offsets are unique) and reject duplicates */
    for (int i = 0; i < count; i++) {
        int lo = 0, hi = count - 1, id = 0;
        while (lo <= hi) {            /* off[] is ascending in file order */
            int mid = (lo + hi) / 2;
            if (off[mid] == o->sorted_off[i]) { id = mid + 1; break; }
            if (off[mid] < o->sorted_off[i]) lo = mid + 1; else hi = mid - 1;
        }
        o->sorted_id[i] = id;
        if (i > 0 && cmp_patterns(&o->sorted_off[i - 1], &o->sorted_off[i]) == 0) {
            free(off); oracle_free(o); return ORACLE_INTERNAL_ERROR;  /* duplicate: ref undefined */
        }
    }
    free(off);

    /* ref PFAC.cpp:693-707 + create_PFACTable_spaceDriven (reorder_Table.cpp:256-329):
     * finals are 1..F, initial F+1, fresh internal states from F+2 in sorted
     * insertion order; the LAST byte of a pattern is pushed without a lookup. */
    o->initial_state = count + 1;
    int next_id = o->initial_state + 1;
    int rc = row_push(o, o->initial_state, 0, 0);     /* make sure rows exist ... */
    if (rc) { oracle_free(o); return rc; }
    o->row_n[o->initial_state] = 0;                   /* ... and are empty */
    for (int p = 0; p < count; p++) {
        const unsigned char *s = o->file + o->sorted_off[p];
        int id = o->sorted_id[p];
        int plen = o->pattern_len[id];
        int state = o->initial_state;
        for (int j = 0; j < plen; j++) {
            int ch = s[j];
            if (j == plen - 1) {
                rc = row_push(o, state, ch, id);
                if (rc) { oracle_free(o); return rc; }
            } else {
                int nx = row_lookup(o, state, ch);
                if (nx == ORACLE_TRAP_STATE) {
                    rc = row_push(o, state, ch, next_id);
                    if (rc) { oracle_free(o); return rc; }
                    state = next_id++;
                } else {
                    state = nx;
                }
            }
        }
    }
    o->num_states = next_id;
    /* make row arrays cover every state id */
    if (o->num_states > 0) {
        rc = row_push(o, o->num_states - 1, 0, 0);
        if (rc) { oracle_free(o); return rc; }
        o->row_n[o->num_states - 1]--;
    }
    o->num_leaves = 0;                                            /* ref PFAC.cpp:716-722 */
    for (int i = 1; i <= count; i++)
        if (o->row_n[i] == 0) o->num_leaves++;
    *out = o;
    return ORACLE_SUCCESS;
}

/* ----------------------------------------------------------------- tables */

/* ref PFAC_create2DTable, PFAC/src/PFAC.cpp:345-402: row-major state*256+ch,
 * TRAP-filled, later edges overwrite earlier ones. */
int oracle_build_dense(oracle_pfac_t *o)
{
    if (!o) return ORACLE_PATTERNS_NOT_READY;
    free(o->dense);
    size_t n = (size_t)o->num_states * ORACLE_CHAR_SET;
    o->dense = (int *)malloc(n * sizeof(int));
    if (!o->dense) return ORACLE_ALLOC_FAILED;
    for (size_t i = 0; i < n; i++) o->dense[i] = ORACLE_TRAP_STATE;
    for (int s = 0; s < o->num_states; s++)
        for (int j = 0; j < o->row_n[s]; j++)
            o->dense[(size_t)s * ORACLE_CHAR_SET + o->row[s][j].ch] = o->row[s][j].next_state;
    return ORACLE_SUCCESS;
}

/* ref PFAC_createHashTable, PFAC/src/PFAC.cpp:422-648. */
int oracle_build_hash(oracle_pfac_t *o)
{
    if (!o) return ORACLE_PATTERNS_NOT_READY;
    const int p = 257;                                            /* ref :438-439 */
    free(o->hash_row); free(o->hash_val); free(o->initial_row);
    o->hash_row = NULL; o->hash_val = NULL; o->initial_row = NULL;
    o->hash_row = (oracle_int2_t *)malloc(sizeof(oracle_int2_t) * (size_t)o->num_states);
    if (!o->hash_row) return ORACLE_ALLOC_FAILED;
    long total = 0;
    for (int i = 0; i < o->num_states; i++) {                     /* ref :449-484: bucket ladder */
        int B = o->row_n[i], S;
        if (B == 0) S = 0;
        else if (B == 1) S = 1;
        else if (B <= 2) S = 4;
        else if (B <= 4) S = 16;
        else if (B == 5) S = 32;
        else if (B <= 8) S = 64;
        else if (B <= 11) S = 128;
        else if (B <= 255) S = 256;
        else { free(o->hash_row); o->hash_row = NULL; return ORACLE_INTERNAL_ERROR; }
        if (B == 0) { o->hash_row[i].x = -1; o->hash_row[i].y = -1; }
        else { o->hash_row[i].x = (int)total; o->hash_row[i].y = S - 1; total += S; }
    }
    o->hash_total = total;
    o->hash_val = (oracle_int2_t *)malloc(sizeof(oracle_int2_t) * (size_t)(total ? total : 1));
    if (!o->hash_val) { free(o->hash_row); o->hash_row = NULL; return ORACLE_ALLOC_FAILED; }
    memset(o->hash_val, 0xFF, sizeof(oracle_int2_t) * (size_t)total);   /* ref :496 */
    for (int i = 0; i < o->num_states; i++) {
        int B = o->row_n[i];
        if (B == 0) continue;
        int S = o->hash_row[i].y + 1;
        int offset = o->hash_row[i].x;
        int k = -1;
        if (S == 1 || S == 256) {                                 /* ref :506-519 */
            k = 1;
        } else {                                                  /* ref :520-551: smallest k in 1..256 */
            int used[256];
            for (int kk = 1; kk <= 256 && k < 0; kk++) {
                int ok = 1;
                for (int j = 0; j < S; j++) used[j] = 0;
                for (int j = 0; j < B; j++) {
                    int pos = ((kk * o->row[i][j].ch) % p) % S;
                    if (used[pos]) { ok = 0; break; }
                    used[pos] = 1;
                }
                if (ok) k = kk;
            }
            if (k < 0) {                                          /* ref :543-551 */
                free(o->hash_row); free(o->hash_val);
                o->hash_row = NULL; o->hash_val = NULL;
                return ORACLE_INTERNAL_ERROR;
            }
        }
        for (int j = 0; j < B; j++) {                             /* ref :553-560 */
            int pos = ((k * o->row[i][j].ch) % p) % S;
            o->hash_val[offset + pos].x = o->row[i][j].next_state;
            o->hash_val[offset + pos].y = o->row[i][j].ch;
        }
        o->hash_row[i].y |= (k << 16);                            /* ref :561 */
    }
    /* ref :564-594: 256-entry row of the initial state, read back THROUGH the hash */
    o->initial_row = (int *)malloc(sizeof(int) * ORACLE_CHAR_SET);
    if (!o->initial_row) return ORACLE_ALLOC_FAILED;
    oracle_int2_t r = o->hash_row[o->initial_state];
    for (int c = 0; c < ORACLE_CHAR_SET; c++) {
        if (r.x == -1) { o->initial_row[c] = ORACLE_TRAP_STATE; continue; }
        int sm1 = r.y & 0xFFFF, k = r.y >> 16;
        int pos = ((k * c) % p) & sm1;
        oracle_int2_t v = o->hash_val[r.x + pos];
        o->initial_row[c] = (v.y == c) ? v.x : ORACLE_TRAP_STATE;
    }
    return ORACLE_SUCCESS;
}

/* --------------------------------------------------------------- matchers */

/* one walk from `start`: ref PFAC_CPU.cpp:76-96 */
static inline int walk_dense(const int *T, int nf, int init, const unsigned char *in, size_t n,
                             size_t start)
{
    int state = init, match = 0;
    for (size_t pos = start; pos < n; pos++) {
        state = T[(size_t)state * ORACLE_CHAR_SET + in[pos]];
        if (state == ORACLE_TRAP_STATE) break;
        if (state <= nf) match = state;
    }
    return match;
}

/* one walk with the hashed lookup: ref PFAC_CPU.cpp:121-157 */
static inline int walk_hash(const oracle_int2_t *rowp, const oracle_int2_t *valp, int nf, int init,
                            const unsigned char *in, size_t n, size_t start)
{
    int state = init, match = 0;
    for (size_t pos = start; pos < n; pos++) {
        int c = in[pos];
        oracle_int2_t r = rowp[state];
        if (r.x < 0) break;                                      /* TRAP */
        int sm1 = r.y & 0xFFFF, k = r.y >> 16;
        int hp = ((k * c) % 257) & sm1;
        oracle_int2_t v = valp[r.x + hp];
        if (v.y != c) break;                                     /* TRAP */
        state = v.x;
        if (state <= nf) match = state;
    }
    return match;
}

int oracle_match_dense(const oracle_pfac_t *o, const unsigned char *in, size_t n, int *out)
{
    if (!o || !o->dense) return ORACLE_PATTERNS_NOT_READY;
    if (o->num_patterns >= o->initial_state) return ORACLE_INTERNAL_ERROR;   /* ref PFAC_CPU.cpp:45-47 */
    for (size_t i = 0; i < n; i++) out[i] = 0;                               /* ref :73-75 */
    for (size_t s = 0; s < n; s++) {
        int m = walk_dense(o->dense, o->num_patterns, o->initial_state, in, n, s);
        if (m) out[s] = m;
    }
    return ORACLE_SUCCESS;
}

int oracle_match_hash(const oracle_pfac_t *o, const unsigned char *in, size_t n, int *out)
{
    if (!o || !o->hash_row) return ORACLE_PATTERNS_NOT_READY;
    if (o->num_patterns >= o->initial_state) return ORACLE_INTERNAL_ERROR;
    for (size_t i = 0; i < n; i++) out[i] = 0;
    for (size_t s = 0; s < n; s++) {
        int m = walk_hash(o->hash_row, o->hash_val, o->num_patterns, o->initial_state, in, n, s);
        if (m) out[s] = m;
    }
    return ORACLE_SUCCESS;
}

int oracle_omp_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ref PFAC_CPU_OMP_timeDriven, PFAC_CPU_OMP.cpp:81-120: serial zero fill,
 * then `#pragma omp parallel for` (default static schedule) over start. */
int oracle_match_dense_omp(const oracle_pfac_t *o, const unsigned char *in, size_t n, int *out,
                           int nthreads)
{
    if (!o || !o->dense) return ORACLE_PATTERNS_NOT_READY;
    if (o->num_patterns >= o->initial_state) return ORACLE_INTERNAL_ERROR;
    const int *T = o->dense;
    const int nf = o->num_patterns, init = o->initial_state;
    for (size_t i = 0; i < n; i++) out[i] = 0;
    long long nn = (long long)n;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for
    for (long long s = 0; s < nn; s++) {
        int m = walk_dense(T, nf, init, in, n, (size_t)s);
        if (m) out[s] = m;
    }
    return ORACLE_SUCCESS;
}

/* ref PFAC_CPU_OMP_spaceDriven, PFAC_CPU_OMP.cpp:123-185 */
int oracle_match_hash_omp(const oracle_pfac_t *o, const unsigned char *in, size_t n, int *out,
                          int nthreads)
{
    if (!o || !o->hash_row) return ORACLE_PATTERNS_NOT_READY;
    if (o->num_patterns >= o->initial_state) return ORACLE_INTERNAL_ERROR;
    const oracle_int2_t *rowp = o->hash_row, *valp = o->hash_val;
    const int nf = o->num_patterns, init = o->initial_state;
    for (size_t i = 0; i < n; i++) out[i] = 0;
    long long nn = (long long)n;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for
    for (long long s = 0; s < nn; s++) {
        int m = walk_hash(rowp, valp, nf, init, in, n, (size_t)s);
        if (m) out[s] = m;
    }
    return ORACLE_SUCCESS;
}

/* ------------------------------------------------------------------- dump */

/* ref printString, PFAC_reorder_Table.cpp:93-105 */
static void print_string(const unsigned char *s, int n, FILE *fp)
{
    fprintf(fp, "%c", '\"');
    for (int i = 0; i < n; i++) {
        int ch = s[i];
        if (ch >= 32 && ch <= 126) fprintf(fp, "%c", ch);
        else fprintf(fp, "%2.2x", ch);
    }
    fprintf(fp, "%c", '\"');
}

/* ref PFAC_dumpTransitionTable, PFAC.cpp:1188-1246 (format strings kept: the
 * dump is a user-visible text format, user guide r1.2 p.21) */
int oracle_dump_table(const oracle_pfac_t *o, FILE *fp)
{
    if (!o) return ORACLE_INVALID_PARAMETER;
    if (!fp) fp = stdout;
    fprintf(fp, "# Transition table: number of states = %d, initial state = %d\n", o->num_states,
            o->initial_state);
    fprintf(fp, "# (current state, input character) -> next state \n");
    for (int s = 0; s < o->num_states; s++) {
        for (int j = 0; j < o->row_n[s]; j++) {
            int ch = o->row[s][j].ch, nx = o->row[s][j].next_state;
            if (nx == ORACLE_TRAP_STATE) continue;
            if (ch >= 32 && ch <= 126) fprintf(fp, "(%4d,%4c) -> %d \n", s, ch, nx);
            else fprintf(fp, "(%4d,%4.2x) -> %d \n", s, ch, nx);
        }
    }
    fprintf(fp, "# Output table: number of final states = %d\n", o->num_patterns);
    fprintf(fp, "# [final state] [matched pattern ID] [pattern length] [pattern(string literal)] \n");
    for (int s = 1; s <= o->num_patterns; s++) {
        fprintf(fp, "%5d %5d %5d    ", s, s, o->pattern_len[s]);
        print_string(o->file + o->pattern_off[s], o->pattern_len[s], fp);
        fprintf(fp, "\n");
    }
    return ORACLE_SUCCESS;
}

int oracle_dump_table_to_path(const oracle_pfac_t *o, const char *path)
{
    FILE *fp = fopen(path, "w");
    if (!fp) return ORACLE_FILE_OPEN_ERROR;
    int rc = oracle_dump_table(o, fp);
    fclose(fp);
    return rc;
}

/* ----------------------------------------------------------------- helpers */

long oracle_reduce(int *result, int *pos, size_t n)
{
    long z = 0;
    for (size_t i = 0; i < n; i++) {
        int m = result[i];
        if (m > 0) { result[z] = m; pos[z] = (int)i; z++; }
    }
    return z;
}

void oracle_digest(const int *result, size_t n, unsigned long long *fnv, unsigned long long *count)
{
    unsigned long long h = 0xcbf29ce484222325ULL, c = 0;
    const unsigned char *b = (const unsigned char *)result;
    for (size_t i = 0; i < n; i++) {
        if (result[i]) c++;
        for (int j = 0; j < 4; j++) { h ^= b[4 * i + j]; h *= 0x100000001b3ULL; }
    }
    if (fnv) *fnv = h;
    if (count) *count = c;
}

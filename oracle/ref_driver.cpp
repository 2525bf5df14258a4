/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin C-ABI shim around the REAL reference host code, compiled unmodified
 * from /root/reference/PFAC/src/{PFAC_reorder_Table,PFAC_CPU,PFAC_CPU_OMP}.cpp
 * into oracle/_ref/libpfac_ref.so (recipe: oracle/Makefile, target `ref`).
 * Those three files need only the type declarations of <cuda_runtime.h>
 * (int2); the image ships a genuine copy of that header inside Triton's
 * NVIDIA backend, so no stand-in header is written.  PFAC.cpp (table
 * materialisers, API) is NOT buildable here: it needs libcudart and a
 * libpfac_sm_<NN>.so kernel module, neither of which exists in the image --
 * see DESIGN.md "oracle".
 *
 * What this exposes to tests/bench:
 *   ref_build()        the reference's parsePatternFile +
 *                      create_PFACTable_spaceDriven, called exactly as
 *                      PFAC_readPatternFromFile does (PFAC.cpp:674-707)
 *   ref_match_dense()  PFAC_CPU_timeDriven / PFAC_CPU_OMP_timeDriven
 *   ref_match_hash()   PFAC_CPU_spaceDriven / PFAC_CPU_OMP_spaceDriven
 * The tables handed to the matchers come from oracle/pfac_oracle.c.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "PFAC_P.h"   /* reference private header: TableEle, int2, parsePatternFile */

/* prototypes of reference functions that have no header declaration */
PFAC_status_t create_PFACTable_spaceDriven(const char **rowPtr, const int *patternLen_table,
                                           const int *patternID_table, const int max_state_num,
                                           const int pattern_num, const int initial_state,
                                           const int baseOfUsableStateID, int *state_num_ptr,
                                           vector<vector<TableEle> > &PFAC_table);
PFAC_status_t PFAC_CPU_timeDriven(char *input_string, const int input_size, int *PFAC_table,
                                  int num_finalState, int initial_state, int *match_result);
PFAC_status_t PFAC_CPU_spaceDriven(char *input_string, const int input_size, int2 *hashRowPtr,
                                   int2 *hashValPtr, const int hash_p, int num_finalState,
                                   int initial_state, int *match_result);
PFAC_status_t PFAC_CPU_OMP_timeDriven(char *input_string, int input_size, int *PFAC_table,
                                      int num_finalState, int initial_state, int *match_result);
PFAC_status_t PFAC_CPU_OMP_spaceDriven(char *input_string, int input_size, int2 *hashRowPtr,
                                       int2 *hashValPtr, int hash_p, int num_finalState,
                                       int initial_state, int *match_result);

extern "C" {

typedef struct {
    int num_patterns;
    int num_states;
    int initial_state;
    int max_pattern_len;
    int num_edges;
    int *edge_state;   /* [num_edges] source state, ascending, insertion order inside a state */
    int *edge_ch;
    int *edge_next;
    int *pattern_len;  /* [num_patterns+1] by ID */
    int *sorted_id;    /* [num_patterns] pattern IDs in the reference's sorted order */
} ref_trie_t;

void ref_free(ref_trie_t *t)
{
    if (!t) return;
    free(t->edge_state); free(t->edge_ch); free(t->edge_next);
    free(t->pattern_len); free(t->sorted_id);
    free(t);
}

int ref_build(const char *pattern_file, ref_trie_t **out)
{
    *out = NULL;
    char **rowPtr = NULL; char *valPtr = NULL;
    int *idTable = NULL, *lenTable = NULL;
    int maxStates = 0, numPatterns = 0;
    PFAC_status_t st = parsePatternFile((char *)pattern_file, &rowPtr, &valPtr, &idTable, &lenTable,
                                        &maxStates, &numPatterns);
    if (PFAC_STATUS_SUCCESS != st) return (int)st;
    int initial = numPatterns + 1;                       /* PFAC.cpp:693 */
    vector<vector<TableEle> > table;
    int numStates = 0;
    st = create_PFACTable_spaceDriven((const char **)rowPtr, lenTable, idTable, maxStates,
                                      numPatterns, initial, initial + 1, &numStates, table);
    if (PFAC_STATUS_SUCCESS != st) { free(rowPtr); free(valPtr); free(idTable); free(lenTable); return (int)st; }

    ref_trie_t *t = (ref_trie_t *)calloc(1, sizeof(*t));
    t->num_patterns = numPatterns;
    t->num_states = numStates;
    t->initial_state = initial;
    int ne = 0;
    for (int s = 0; s < numStates; s++) ne += (int)table[s].size();
    t->num_edges = ne;
    t->edge_state = (int *)malloc(sizeof(int) * (ne ? ne : 1));
    t->edge_ch = (int *)malloc(sizeof(int) * (ne ? ne : 1));
    t->edge_next = (int *)malloc(sizeof(int) * (ne ? ne : 1));
    int e = 0;
    for (int s = 0; s < numStates; s++)
        for (size_t j = 0; j < table[s].size(); j++) {
            t->edge_state[e] = s; t->edge_ch[e] = table[s][j].ch; t->edge_next[e] = table[s][j].nextState; e++;
        }
    t->pattern_len = (int *)malloc(sizeof(int) * (numPatterns + 1));
    t->sorted_id = (int *)malloc(sizeof(int) * (numPatterns ? numPatterns : 1));
    t->max_pattern_len = 0;
    for (int i = 0; i <= numPatterns; i++) {
        t->pattern_len[i] = lenTable[i];
        if (i > 0 && lenTable[i] > t->max_pattern_len) t->max_pattern_len = lenTable[i];
    }
    for (int i = 0; i < numPatterns; i++) t->sorted_id[i] = idTable[i];
    free(rowPtr); free(valPtr); free(idTable); free(lenTable);
    *out = t;
    return 0;
}

int ref_match_dense(char *in, int n, int *table, int num_final, int initial, int *out, int use_omp)
{
    return (int)(use_omp ? PFAC_CPU_OMP_timeDriven(in, n, table, num_final, initial, out)
                         : PFAC_CPU_timeDriven(in, n, table, num_final, initial, out));
}

int ref_match_hash(char *in, int n, int *row_ptr, int *val_ptr, int num_final, int initial, int *out,
                   int use_omp)
{
    return (int)(use_omp ? PFAC_CPU_OMP_spaceDriven(in, n, (int2 *)row_ptr, (int2 *)val_ptr, 257,
                                                    num_final, initial, out)
                         : PFAC_CPU_spaceDriven(in, n, (int2 *)row_ptr, (int2 *)val_ptr, 257,
                                                num_final, initial, out));
}

} /* extern "C" */

/*
 * cpu_engine.cpp -- PFAC_PLATFORM_CPU and PFAC_PLATFORM_CPU_OMP for
 * PFAC_matchFromHost (an explicit user choice through PFAC_setPlatform; the
 * GPU platform never falls back to this code).
 *
 * Semantics follow PFAC/src/PFAC_CPU.cpp:43-163 and
 * PFAC/src/PFAC_CPU_OMP.cpp:65-185: one walk per start position through the
 * table of the current perfMode, remembering the last final state seen.
 * Unlike the reference the result is written once per position (no separate
 * serial zero-fill pass) and sizes are size_t.
 */
#include "pfac_host.h"

namespace pfac {

namespace {

struct DenseStep {
    const int *table;
    inline int operator()(int state, int ch) const { return table[(size_t)state * kCharSet + ch]; }
};

struct HashStep {
    const Int2 *rowPtr;
    const Int2 *valPtr;
    inline int operator()(int state, int ch) const
    {
        const Int2 r = rowPtr[state];
        if (r.x < 0) return kTrapState;
        const int slot = (((r.y >> 16) * ch) % kHashP) & (r.y & 0xFFFF);
        const Int2 v = valPtr[r.x + slot];
        return v.y == ch ? v.x : kTrapState;
    }
};

template <class Step>
void scan(const Step &step, int numFinal, int initial, const unsigned char *in, size_t n, int *out,
          bool useOpenMP)
{
    const long long nn = (long long)n;
#pragma omp parallel for schedule(static) if (useOpenMP)
    for (long long start = 0; start < nn; start++) {
        int state = initial, match = 0;
        for (size_t pos = (size_t)start; pos < n; pos++) {
            state = step(state, in[pos]);
            if (state == kTrapState) break;
            if (state <= numFinal) match = state;
        }
        out[start] = match;
    }
}

} // namespace

PFAC_status_t matchOnCpu(const PFAC_context *ctx, const unsigned char *in, size_t n, int *out,
                         bool useOpenMP)
{
    const Automaton &fa = ctx->fa;
    if (fa.numPatterns >= fa.initialState) return PFAC_STATUS_INTERNAL_ERROR;   /* ref PFAC_CPU.cpp:45-47 */
    if (ctx->perfMode == PFAC_TIME_DRIVEN) {
        if (ctx->h_dense.empty()) return PFAC_STATUS_INTERNAL_ERROR;
        scan(DenseStep{ctx->h_dense.data()}, fa.numPatterns, fa.initialState, in, n, out, useOpenMP);
    } else {
        if (ctx->h_hashRow.empty()) return PFAC_STATUS_INTERNAL_ERROR;
        scan(HashStep{ctx->h_hashRow.data(), ctx->h_hashVal.data()}, fa.numPatterns, fa.initialState,
             in, n, out, useOpenMP);
    }
    return PFAC_STATUS_SUCCESS;
}

} // namespace pfac

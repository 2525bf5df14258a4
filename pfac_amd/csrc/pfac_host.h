/*
 * pfac_host.h -- internal declarations of the host side of libpfac.so.
 */
#ifndef PFAC_HOST_H_
#define PFAC_HOST_H_

#include "pfac_context.h"

namespace pfac {

/* pattern_compiler.cpp */
PFAC_status_t compilePatternFile(const char *filename, Automaton &fa, unsigned int flags = 0);
PFAC_status_t compilePatternBytes(std::vector<unsigned char> bytes, Automaton &fa, unsigned int flags = 0);   /* flags: PFACX_READ_* */
void buildInitialRow(const Automaton &fa, std::vector<int> &row);
void buildFilter(const Automaton &fa, Filter &f);
void buildReduceFilter(const Automaton &fa, Filter &f);     /* gram1 + prefix4 alone (buildFilter calls it; a compiled set is loaded without them) */

/* tables.cpp */
PFAC_status_t buildDenseTable(const Automaton &fa, std::vector<int> &dense);
PFAC_status_t buildHashTable(const Automaton &fa, std::vector<Int2> &rowPtr,
                             std::vector<Int2> &valPtr);

PFAC_status_t buildChainedHashTable(const Automaton &fa, std::vector<ChainSlot> &slots, int &jumpLog2);

/* cpu_engine.cpp: PFAC_PLATFORM_CPU / PFAC_PLATFORM_CPU_OMP */
PFAC_status_t matchOnCpu(const PFAC_context *ctx, const unsigned char *in, size_t n, int *out,
                         bool useOpenMP);

} // namespace pfac

#endif

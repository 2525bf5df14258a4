/*
 * compiled_set.cpp -- PFACX_saveCompiled / PFACX_loadCompiled (include/pfac_ext.h): a pattern set on disk in parsed form.
 * Nothing a kernel indexes memory with is taken from a file: the trie is checked to be one, every table and every prefilter
 * bitmap is rebuilt from it.
 */
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "pfac_host.h"

using pfac::Int2;
using namespace pfac_internal;

namespace {

constexpr char kCompiledMagic[8] = {'P', 'F', 'A', 'C', 'X', 'C', '1', 0};
constexpr uint32_t kCompiledVersion = 7;          /* 2: patterns of 1-2 bytes are folded into the 3-gram bitmap; 3: root bucket + jump table behind the chained slots;
                                                     5: the prefix ladder replaces the 4-gram bitmap, the chained table has its own compact breadth-first layout;
                                                     6: 36-byte walk-queue entries (layout fingerprint); 7: no transition table is stored any more -- hashed and
                                                     chained tables are rebuilt from the checked trie at load (a file cannot steer a device read) --, the scalars
                                                     carry the pattern file's ignored trailing bytes */
/* what the stored tables depend on besides the patterns: hash constants and slot layout */
constexpr uint32_t kLayoutFingerprint = pfac::kGram3Mul ^ (pfac::kLadMul0 * 3u) ^ (pfac::kLadMul * 5u) ^ (pfac::kLadMulS * 11u) ^ (pfac::kLadMulG * 13u) ^ (pfac::kLadMulG2 * 17u) ^ (pfac::kFinal3Mul * 7u) ^ (pfac::kFinal3Mul2 * 19u) ^
                                        ((uint32_t)pfac::kLadderLevels << 12) ^
                                        ((uint32_t)sizeof(pfac::ChainSlot) << 24) ^ ((uint32_t)pfac::kChainMax << 20) ^ 0x20u /* entry bytes */ ^
                                        0x4000u /* chained table: multiply-shift bucket hash */;
struct CompiledHeader {
    char magic[8];
    uint32_t version, fingerprint, perfMode, jumpLog2;   /* jumpLog2: log2 of the jump-table slots at the end of the chained table */
    uint64_t payloadBytes, payloadFnv1a;
};
enum Section : uint32_t { kSecFile = 1, kSecScalars, kSecPatOff, kSecPatLen, kSecSorted, kSecEdgeBegin, kSecEdgeCh, kSecEdgeNext,
                          kSecFilter, kSecGram3, kSecLadder, kSecFinal3, kSecShort, kSecHashRow, kSecHashVal, kSecChain, kSecRootUnused, kSecInitialRow };

uint64_t fnv1a64(const unsigned char *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

template <class T>
void putSection(std::vector<unsigned char> &out, uint32_t tag, const T *data, size_t count)
{
    const uint64_t bytes = (uint64_t)count * sizeof(T);
    const unsigned char *t = reinterpret_cast<const unsigned char *>(&tag), *b = reinterpret_cast<const unsigned char *>(&bytes);
    out.insert(out.end(), t, t + 4);
    out.insert(out.end(), b, b + 8);
    const unsigned char *d = reinterpret_cast<const unsigned char *>(data);
    out.insert(out.end(), d, d + bytes);
}

template <class T>
bool takeSection(const unsigned char *p, uint64_t bytes, std::vector<T> &v)
{
    if (bytes % sizeof(T)) return false;
    v.resize(bytes / sizeof(T));
    if (bytes) std::memcpy(v.data(), p, bytes);
    return true;
}

} // namespace

extern "C" {

PFAC_status_t PFACX_saveCompiled(PFAC_handle_t handle, const char *filename)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!filename) return PFAC_STATUS_INVALID_PARAMETER;
    if (!handle->isPatternsReady) return PFAC_STATUS_PATTERNS_NOT_READY;
    std::lock_guard<std::mutex> guard(handle->lock);
    PFAC_context *c = handle;
    try {
        const pfac::Automaton &fa = c->fa;
        const pfac::Filter &f = c->filter;
        std::vector<unsigned char> payload;
        putSection(payload, kSecFile, fa.file.data(), fa.file.size());
        const int64_t scalars[6] = {fa.numPatterns, fa.maxPatternLen, fa.initialState, fa.numStates, fa.numLeaves, (int64_t)fa.trailingBytes};
        putSection(payload, kSecScalars, scalars, 6);
        putSection(payload, kSecPatOff, fa.patternOff.data(), fa.patternOff.size());
        putSection(payload, kSecPatLen, fa.patternLen.data(), fa.patternLen.size());
        putSection(payload, kSecSorted, fa.sortedId.data(), fa.sortedId.size());
        putSection(payload, kSecEdgeBegin, fa.edgeBegin.data(), fa.edgeBegin.size());
        putSection(payload, kSecEdgeCh, fa.edgeCh.data(), fa.edgeCh.size());
        putSection(payload, kSecEdgeNext, fa.edgeNext.data(), fa.edgeNext.size());
        const uint64_t filt[10] = {(uint64_t)f.log2Bits, (uint64_t)f.log2BitsLad, (uint64_t)f.log2BitsF3, f.hasShort ? 1u : 0u, f.bitsSet, f.bitsSetLad,
                                  f.ladderStops, f.ladderGoOns, (uint64_t)f.ladderThin, (uint64_t)f.ladderExtend};
        putSection(payload, kSecFilter, filt, 10);
        putSection(payload, kSecGram3, f.gram3.data(), f.gram3.size());
        putSection(payload, kSecLadder, f.ladder.data(), f.ladder.size());
        putSection(payload, kSecFinal3, f.final3.data(), f.final3.size());
        putSection(payload, kSecShort, f.shortBits.data(), f.shortBits.size());
        /* no transition table: dense, hashed and chained tables and the initial row are rebuilt from the edges at load */
        CompiledHeader h;
        std::memset(&h, 0, sizeof(h));
        std::memcpy(h.magic, kCompiledMagic, 8);
        h.version = kCompiledVersion; h.fingerprint = kLayoutFingerprint; h.perfMode = (uint32_t)c->perfMode;
        h.jumpLog2 = 0;                                        /* (was: log2 of the stored chained table's jump slots) */
        h.payloadBytes = payload.size(); h.payloadFnv1a = fnv1a64(payload.data(), payload.size());
        FILE *fp = std::fopen(filename, "wb");
        if (!fp) return PFAC_STATUS_FILE_OPEN_ERROR;
        const bool ok = std::fwrite(&h, sizeof(h), 1, fp) == 1 && (payload.empty() || std::fwrite(payload.data(), payload.size(), 1, fp) == 1);
        return (std::fclose(fp) == 0 && ok) ? PFAC_STATUS_SUCCESS : PFAC_STATUS_INTERNAL_ERROR;
    } catch (const std::bad_alloc &) { return PFAC_STATUS_ALLOC_FAILED; }
}

PFAC_status_t PFACX_loadCompiled(PFAC_handle_t handle, const char *filename)
{
    if (!handle) return PFAC_STATUS_INVALID_HANDLE;
    if (!filename) return PFAC_STATUS_INVALID_PARAMETER;
    FILE *fp = std::fopen(filename, "rb");
    if (!fp) return PFAC_STATUS_FILE_OPEN_ERROR;
    CompiledHeader h;
    std::vector<unsigned char> payload;
    bool ok = std::fread(&h, sizeof(h), 1, fp) == 1 && std::memcmp(h.magic, kCompiledMagic, 8) == 0 && h.version == kCompiledVersion &&
              h.fingerprint == kLayoutFingerprint && (h.perfMode == PFAC_TIME_DRIVEN || h.perfMode == PFAC_SPACE_DRIVEN) &&
              h.payloadBytes < (uint64_t(1) << 40);
    if (ok) {
        /* the payload is what the file holds behind the header, to the byte: a header that announces more must not make the loader
         * ask for a terabyte (tools/fuzz_host.cpp, round 5: found after 20 000 mutations) */
        long size = -1;
        if (std::fseek(fp, 0, SEEK_END) == 0) size = std::ftell(fp);
        ok = size >= (long)sizeof(h) && (uint64_t)(size - (long)sizeof(h)) == h.payloadBytes && std::fseek(fp, (long)sizeof(h), SEEK_SET) == 0;
    }
    try {
        if (ok) {
            payload.resize((size_t)h.payloadBytes);
            ok = payload.empty() || std::fread(payload.data(), payload.size(), 1, fp) == 1;
        }
    } catch (const std::bad_alloc &) { std::fclose(fp); return PFAC_STATUS_ALLOC_FAILED; }
    std::fclose(fp);
    if (!ok || fnv1a64(payload.data(), payload.size()) != h.payloadFnv1a) return PFAC_STATUS_INVALID_PARAMETER;   /* not a compiled set of this build, or damaged */

    /* everything is parsed and checked in temporaries: a refused file leaves the handle as it was.  The checksum is no
     * protection against a crafted file (FNV-1a is recomputed in a line), so nothing that a kernel indexes memory with is
     * taken from the file: the trie is checked to BE a trie of the stored patterns' depth, and every transition table --
     * hashed, chained, the initial row -- is rebuilt from it.  The prefilter bitmaps are taken as they are: they are read
     * with masked LDS addresses, a wrong bit can cost a result, not a memory access. */
    pfac::Automaton fa;
    pfac::Filter f;
    std::vector<int64_t> scalars;
    std::vector<uint64_t> filt;
    try {
        size_t at = 0;
        while (ok && at + 12 <= payload.size()) {
            uint32_t tag; uint64_t bytes;
            std::memcpy(&tag, &payload[at], 4); std::memcpy(&bytes, &payload[at + 4], 8);
            at += 12;
            if (bytes > payload.size() - at) { ok = false; break; }
            const unsigned char *p = payload.data() + at;
            switch (tag) {
            case kSecFile: ok = takeSection(p, bytes, fa.file); break;
            case kSecScalars: ok = takeSection(p, bytes, scalars); break;
            case kSecPatOff: ok = takeSection(p, bytes, fa.patternOff); break;
            case kSecPatLen: ok = takeSection(p, bytes, fa.patternLen); break;
            case kSecSorted: ok = takeSection(p, bytes, fa.sortedId); break;
            case kSecEdgeBegin: ok = takeSection(p, bytes, fa.edgeBegin); break;
            case kSecEdgeCh: ok = takeSection(p, bytes, fa.edgeCh); break;
            case kSecEdgeNext: ok = takeSection(p, bytes, fa.edgeNext); break;
            case kSecFilter: ok = takeSection(p, bytes, filt); break;
            case kSecGram3: ok = takeSection(p, bytes, f.gram3); break;
            case kSecLadder: ok = takeSection(p, bytes, f.ladder); break;
            case kSecFinal3: ok = takeSection(p, bytes, f.final3); break;
            case kSecShort: ok = takeSection(p, bytes, f.shortBits); break;
            default: break;                                    /* unknown section of a later writer: skipped */
            }
            at += (size_t)bytes;
        }
        ok = ok && scalars.size() == 6 && filt.size() == 10;
        for (size_t i = 0; ok && i < scalars.size(); i++) ok = scalars[i] >= 0 && scalars[i] < (int64_t(1) << 31);
        if (ok) {
            fa.numPatterns = (int)scalars[0]; fa.maxPatternLen = (int)scalars[1]; fa.initialState = (int)scalars[2];
            fa.numStates = (int)scalars[3]; fa.numLeaves = (int)scalars[4]; fa.trailingBytes = (size_t)scalars[5];
            f.log2Bits = (int)filt[0]; f.log2BitsLad = (int)filt[1]; f.log2BitsF3 = (int)filt[2]; f.hasShort = filt[3] != 0;
            f.bitsSet = (size_t)filt[4]; f.bitsSetLad = (size_t)filt[5];
            f.ladderStops = (size_t)filt[6]; f.ladderGoOns = (size_t)filt[7]; f.ladderThin = (int)filt[8]; f.ladderExtend = (int)filt[9];
            const size_t S = (size_t)fa.numStates, F = (size_t)fa.numPatterns;
            ok = fa.numStates > 0 && fa.initialState == fa.numPatterns + 1 && (size_t)fa.initialState < S &&
                 fa.patternOff.size() == F + 1 && fa.patternLen.size() == F + 1 && fa.sortedId.size() == F &&
                 fa.edgeBegin.size() == S + 1 && fa.edgeCh.size() == fa.edgeNext.size() && !fa.edgeBegin.empty() &&
                 fa.edgeBegin[0] == 0 && (size_t)fa.edgeBegin.back() == fa.edgeCh.size() && fa.trailingBytes <= fa.file.size() &&
                 f.log2Bits >= 13 && f.log2Bits <= 18 && f.log2BitsLad >= 13 && f.log2BitsLad <= 19 && f.log2BitsF3 >= 10 && f.log2BitsF3 <= 13 &&
                 pfac::kGram3LdsBytes + ((size_t(1) << f.log2BitsLad) + (size_t(1) << f.log2BitsF3)) / 8 + (f.hasShort ? 8192u : 0u) <= pfac::kFilterLdsBudget &&
                 f.gram3.size() == (size_t(1) << f.log2Bits) / 32 && f.ladder.size() == (size_t(1) << f.log2BitsLad) / 32 &&
                 f.final3.size() == (size_t(1) << f.log2BitsF3) / 32 && f.shortBits.size() == 65536 / 32;
            for (size_t i = 0; ok && i + 1 < fa.edgeBegin.size(); i++) ok = fa.edgeBegin[i] <= fa.edgeBegin[i + 1] && fa.edgeBegin[i] >= 0 && fa.edgeBegin[i + 1] - fa.edgeBegin[i] <= pfac::kCharSet;
            for (size_t i = 0; ok && i < fa.edgeNext.size(); i++) ok = fa.edgeNext[i] > 0 && (size_t)fa.edgeNext[i] < S && fa.edgeNext[i] != fa.initialState;
            /* what the kernels and the host path take on trust: the longest pattern (overlap of pieces and slices, the
             * safety margin at the end of the input), the pattern lengths and offsets */
            int longest = 0;
            for (size_t id = 1; ok && id <= F; id++) {
                ok = fa.patternLen[id] >= 1 && fa.patternOff[id] >= 0 && (size_t)fa.patternOff[id] + (size_t)fa.patternLen[id] <= fa.file.size();
                longest = fa.patternLen[id] > longest ? fa.patternLen[id] : longest;
            }
            ok = ok && fa.maxPatternLen == longest;
            /* the edges form a TREE below the initial state: every state is entered by at most one edge, the bytes of a
             * state's edges are distinct, no state lies deeper than the longest pattern (so no walk is longer: a cycle would
             * take a walk past the margin the kernels keep at the end of the input), and final state `id` lies exactly
             * patternLen[id] deep */
            if (ok) {
                std::vector<int> depth(S, -1);
                std::vector<int> order;
                order.reserve(S);
                depth[(size_t)fa.initialState] = 0;
                order.push_back(fa.initialState);
                for (size_t at2 = 0; ok && at2 < order.size(); at2++) {
                    const int st = order[at2];
                    uint64_t seen[4] = {0, 0, 0, 0};
                    for (int e = fa.edgeBegin[(size_t)st]; ok && e < fa.edgeBegin[(size_t)st + 1]; e++) {
                        const unsigned ch = fa.edgeCh[(size_t)e];
                        const int nx = fa.edgeNext[(size_t)e];
                        ok = !(seen[ch >> 6] & (uint64_t(1) << (ch & 63))) && depth[(size_t)nx] < 0 && depth[(size_t)st] < fa.maxPatternLen;
                        seen[ch >> 6] |= uint64_t(1) << (ch & 63);
                        if (ok) { depth[(size_t)nx] = depth[(size_t)st] + 1; order.push_back(nx); }
                    }
                }
                for (size_t id = 1; ok && id <= F; id++) ok = depth[id] == fa.patternLen[id];
                /* states the initial state does not reach must have no edges (state 0 is the unused one) */
                for (size_t st = 0; ok && st < S; st++) ok = depth[st] >= 0 || fa.edgeBegin[st] == fa.edgeBegin[st + 1];
            }
        }
    } catch (const std::bad_alloc &) { return PFAC_STATUS_ALLOC_FAILED; }
    if (!ok) return PFAC_STATUS_INVALID_PARAMETER;

    std::lock_guard<std::mutex> guard(handle->lock);
    std::unique_lock<std::shared_mutex> tables(handle->tablesInUse);
    PFAC_context *c = handle;
    if (c->isPatternsReady) freeResources(c);
    c->patternFile = filename;
    c->perfMode = (int)h.perfMode;
    c->fa = std::move(fa);
    /* The prefilter bitmaps are rebuilt from the checked trie as well, like every table (30 ms for a Snort-scale set): a stale or
     * crafted file with a valid checksum could not make a kernel read outside a bitmap (addresses are masked), but a cleared
     * bit silently drops matches, and the full-result path (gram3 / ladder from the file) could disagree with the compacted-
     * output path (gram1 / prefix4, always rebuilt).  The file's copies are read, size-checked and dropped. */
    c->filter = pfac::Filter();
    c->isPatternsReady = true;
    pfac::buildInitialRow(c->fa, c->h_initialRow);
    PFAC_status_t st;
    try { st = bindCommon(c, /*build=*/true); } catch (const std::bad_alloc &) { st = PFAC_STATUS_ALLOC_FAILED; }
    if (st == PFAC_STATUS_SUCCESS) st = bindTable(c);
    if (st != PFAC_STATUS_SUCCESS) { freeResources(c); return st; }
    return PFAC_STATUS_SUCCESS;
}

} /* extern "C" */

/*
 * pattern_compiler.cpp -- pattern file -> failureless trie (host side).
 *
 * Behavioural contract = the reference's parsePatternFile
 * (PFAC/src/PFAC_reorder_Table.cpp:121-231), pattern_cmp_functor (:37-72)
 * and create_PFACTable_spaceDriven (:256-329) as driven by
 * PFAC_readPatternFromFile (PFAC/src/PFAC.cpp:674-722):
 *   - one pattern per '\n'-terminated line, IDs 1..F in file order, bytes
 *     after the last '\n' ignored;
 *   - patterns inserted in (signed char, proper-prefix-first) order;
 *   - final states are the pattern IDs 1..F, the initial state is F+1,
 *     internal states are numbered from F+2 in insertion order;
 *   - the last byte of a pattern is appended to its parent WITHOUT a lookup.
 * The state numbering matters for PFAC_dumpTransitionTable and for the
 * layout of the hashed table, both of which are compared byte-for-byte with
 * the oracle.  The data structures are this implementation's own: edges are
 * kept in per-state singly linked lists during construction (O(1) append,
 * insertion-order scan) and flattened to CSR.
 *
 * Inputs the reference leaves undefined are rejected with a status instead:
 *   - a blank line followed by another pattern (reference: assert at
 *     PFAC_reorder_Table.cpp:291)            -> PFAC_STATUS_INVALID_PARAMETER
 * Duplicate patterns (reference: the comparator returns true for equal patterns, which std::sort does
 * not allow; two edges for one byte; hashed build fails, PFAC.cpp:543-551) are accepted as ONE pattern
 * that is reported under the highest of its IDs.
 */
#include <algorithm>
#include <cstdio>
#include <cstring>

#include "pfac_host.h"

namespace pfac {

namespace {

struct PatRef { int off; int len; int id; };

/* signed-char lexicographic, shorter-is-first on a tie (ref :37-72) */
struct PatLess {
    const unsigned char *base;
    bool operator()(const PatRef &a, const PatRef &b) const
    {
        const int n = a.len < b.len ? a.len : b.len;
        const signed char *s = reinterpret_cast<const signed char *>(base + a.off);
        const signed char *t = reinterpret_cast<const signed char *>(base + b.off);
        for (int i = 0; i < n; i++) {
            if (s[i] != t[i]) return s[i] < t[i];
        }
        return a.len < b.len;
    }
};

/* growable trie with per-state linked edge lists */
class TrieBuilder {
public:
    explicit TrieBuilder(int firstStates) { grow(firstStates); }

    int find(int state, int ch) const
    {
        for (int e = head_[state]; e >= 0; e = link_[e])
            if (ch_[e] == ch) return to_[e];
        return kTrapState;
    }
    /* redirect the existing transition (state, ch); returns its previous target */
    int retarget(int state, int ch, int to)
    {
        for (int e = head_[state]; e >= 0; e = link_[e])
            if (ch_[e] == ch) { const int old = to_[e]; to_[e] = to; return old; }
        return kTrapState;
    }
    void append(int state, int ch, int to)
    {
        const int e = (int)ch_.size();
        ch_.push_back((unsigned char)ch);
        to_.push_back(to);
        link_.push_back(-1);
        if (tail_[state] < 0) head_[state] = e; else link_[tail_[state]] = e;
        tail_[state] = e;
        count_[state]++;
    }
    void grow(int states)
    {
        if ((int)head_.size() >= states) return;
        head_.resize(states, -1); tail_.resize(states, -1); count_.resize(states, 0);
    }
    int fanout(int state) const { return count_[state]; }

    void flatten(int numStates, Automaton &fa) const
    {
        fa.edgeBegin.assign((size_t)numStates + 1, 0);
        for (int s = 0; s < numStates; s++) fa.edgeBegin[s + 1] = fa.edgeBegin[s] + count_[s];
        fa.edgeCh.resize(ch_.size());
        fa.edgeNext.resize(ch_.size());
        for (int s = 0; s < numStates; s++) {
            int w = fa.edgeBegin[s];
            for (int e = head_[s]; e >= 0; e = link_[e]) { fa.edgeCh[w] = ch_[e]; fa.edgeNext[w] = to_[e]; w++; }
        }
    }

private:
    std::vector<int> head_, tail_, count_, link_, to_;
    std::vector<unsigned char> ch_;
};

} // namespace

PFAC_status_t compilePatternFile(const char *filename, Automaton &fa, unsigned int flags)
{
    fa = Automaton();
    if (!filename) return PFAC_STATUS_INVALID_PARAMETER;
    FILE *fp = std::fopen(filename, "rb");
    if (!fp) return PFAC_STATUS_FILE_OPEN_ERROR;
    std::fseek(fp, 0, SEEK_END);
    long fsz = std::ftell(fp);
    std::rewind(fp);
    if (fsz < 0) { std::fclose(fp); return PFAC_STATUS_FILE_OPEN_ERROR; }
    std::vector<unsigned char> bytes;
    try {
        bytes.resize((size_t)fsz);
    } catch (...) { std::fclose(fp); return PFAC_STATUS_ALLOC_FAILED; }
    size_t got = fsz ? std::fread(bytes.data(), 1, (size_t)fsz, fp) : 0;
    std::fclose(fp);
    bytes.resize(got);
    return compilePatternBytes(std::move(bytes), fa, flags);
}

/* the pattern-file format from memory: one pattern per '\n'-terminated line (PFAC_reorder_Table.cpp:121-231) */
PFAC_status_t compilePatternBytes(std::vector<unsigned char> bytes, Automaton &fa, unsigned int flags)
{
    fa = Automaton();
    if (flags & PFACX_READ_STRIP_CR) {              /* "\r\n" line ends: the '\r' is not part of the pattern (the reference keeps it, user guide r1.2 p.15 item 5) */
        size_t w = 0;
        for (size_t i = 0; i < bytes.size(); i++)
            if (!(bytes[i] == '\r' && i + 1 < bytes.size() && bytes[i + 1] == '\n')) bytes[w++] = bytes[i];
        bytes.resize(w);
    }
    fa.file = std::move(bytes);
    const size_t got = fa.file.size();
    {   /* bytes behind the last newline: the reference counts newline-terminated lines only and silently drops them
         * (PFAC_reorder_Table.cpp:181-195); so does this build, but it says so (PFACX_getInfo) or refuses (PFACX_READ_STRICT) */
        size_t last = got;
        while (last > 0 && fa.file[last - 1] != '\n') last--;
        fa.trailingBytes = got - last;
        if ((flags & PFACX_READ_STRICT) && fa.trailingBytes) return PFAC_STATUS_INVALID_PARAMETER;
    }

    /* split into lines; `start` only moves past a NON-empty line (ref :181-190),
     * so a blank line poisons the next pattern -- reported, not asserted. */
    std::vector<PatRef> pats;
    const unsigned char *buf = fa.file.data();
    int start = 0, cur = 0;
    for (size_t i = 0; i < got; i++) {
        if (buf[i] != '\n') { cur++; continue; }
        if (i > 0 && buf[i - 1] != '\n') {
            if (buf[start] == '\n') return PFAC_STATUS_INVALID_PARAMETER;
            pats.push_back(PatRef{start, cur, (int)pats.size() + 1});
            start = (int)i + 1;
        }
        cur = 0;
    }

    const int F = (int)pats.size();
    fa.numPatterns = F;
    fa.patternOff.assign((size_t)F + 1, 0);
    fa.patternLen.assign((size_t)F + 1, 0);
    fa.maxPatternLen = 0;
    for (const PatRef &p : pats) {
        fa.patternOff[p.id] = p.off;
        fa.patternLen[p.id] = p.len;
        fa.maxPatternLen = std::max(fa.maxPatternLen, p.len);
    }
    std::sort(pats.begin(), pats.end(), PatLess{buf});
    fa.sortedId.resize(F);
    for (int i = 0; i < F; i++) fa.sortedId[i] = pats[i].id;

    fa.initialState = F + 1;                       /* ref PFAC.cpp:693 */
    int nextId = F + 2;                            /* ref PFAC.cpp:703 */
    TrieBuilder trie(F + 2);
    for (const PatRef &p : pats) {
        int state = fa.initialState;
        for (int j = 0; j < p.len; j++) {
            const int ch = buf[p.off + j];
            if (j == p.len - 1) {
                /* An existing edge here can only come from an identical pattern (they are adjacent in the sorted
                 * order, nothing hangs below the first one yet).  Rule sets do repeat lines: the copies are one
                 * pattern, reported under the HIGHEST of their IDs; the other IDs are never reported.  (The
                 * reference pushes a second edge for the same byte: its dense table then reports the ID the
                 * unstable sort happened to place last and loses every longer pattern that extends the first
                 * copy, and its hashed build fails, PFAC.cpp:376-381, 506-551.) */
                const int prev = trie.find(state, ch);
                if (prev == kTrapState) trie.append(state, ch, p.id);
                else if (prev < p.id) trie.retarget(state, ch, p.id);
            } else {
                int nx = trie.find(state, ch);
                if (nx == kTrapState) {
                    nx = nextId++;
                    trie.grow(nextId);
                    trie.append(state, ch, nx);
                }
                state = nx;
            }
        }
    }
    fa.numStates = nextId;
    trie.grow(nextId);
    fa.numLeaves = 0;                              /* ref PFAC.cpp:716-722 */
    for (int s = 1; s <= F; s++)
        if (trie.fanout(s) == 0) fa.numLeaves++;
    trie.flatten(fa.numStates, fa);
    return PFAC_STATUS_SUCCESS;
}

/* 256-entry transition row of the initial state: what the reference keeps in
 * shared memory as phi_s02s1 (PFAC_kernel.cu:398-403) and, for the hashed
 * mode, in d_tableOfInitialState (PFAC.cpp:564-594). */
void buildInitialRow(const Automaton &fa, std::vector<int> &row)
{
    row.assign(kCharSet, kTrapState);
    const int s = fa.initialState;
    for (int e = fa.edgeBegin[s]; e < fa.edgeBegin[s + 1]; e++) row[fa.edgeCh[e]] = fa.edgeNext[e];
}

/*
 * Prefilter bitmaps (this implementation only; struct Filter in pfac_context.h has the contract).
 *
 * A start position j can report a non-zero pattern only if
 *   (a) a pattern of length 1 or 2 matches at j                -> shortBits (exact over c0,c1;
 *       a 1-byte pattern sets all 256 c1 slots of its c0)
 *   (b) the walk from j survives three transitions             -> gram3
 * because every match of length >= 3 passes through a depth-3 state; and, deeper, only if
 *   (c) (a), or a pattern of length exactly 3 matches at j     -> final3
 *   (d) or the walk reaches an S node of the prefix ladder through G nodes -> ladder.
 * The kernel tests (b) from LDS for every position and (a)/(c)/(d) for the survivors; only positions
 * that pass are walked through the real table, so false positives cost time, never correctness.
 */
static int sizeLog2(size_t keys, int lo, int hi)
{
    int lg = lo;
    while (lg < hi && (size_t(1) << lg) < keys * 128) lg++;
    return lg;
}

namespace {

/* The ladder of one pattern set for a given "thin" threshold: visits every ladder node that a candidate can reach
 * through G nodes, as S or G, with the rolling hash of its prefix. */
struct LadderWalk {
    const Automaton &fa;
    const std::vector<uint32_t> &below;        /* patterns in the subtree of each state (itself included) */
    uint32_t thin;
    int extend = 0;                            /* thin nodes go on for this many more levels before they stop */
    int last = kLadderLast;                    /* deepest level: its nodes are all stops */
    uint32_t salt = 0;                         /* pfac::Filter::ladderSalt */
    template <class Visit> void run(Visit &&visit) const
    {
        struct Item { int state; int depth; uint32_t acc; int ext; };   /* acc: bytes so far (depth < 4: the bytes; depth >= 4: rolling hash) */
        std::vector<Item> stack;
        stack.push_back({fa.initialState, 0, 0u, extend});
        const int F = fa.numPatterns;
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            if (it.depth < kLadderFirst) {                     /* below the first level: every path, finals on the way included */
                for (int e = fa.edgeBegin[it.state]; e < fa.edgeBegin[it.state + 1]; e++) {
                    const uint32_t acc = it.acc | ((uint32_t)fa.edgeCh[e] << (8 * it.depth));
                    stack.push_back({fa.edgeNext[e], it.depth + 1, it.depth + 1 == kLadderFirst ? ladderStart(acc, salt) : acc, it.ext});
                }
                continue;
            }
            /* a ladder node */
            bool endsSoon = it.state <= F || it.depth >= last;
            for (int e = fa.edgeBegin[it.state]; e < fa.edgeBegin[it.state + 1] && !endsSoon; e++) endsSoon = fa.edgeNext[e] <= F;
            const bool isThin = below[it.state] <= thin;
            const bool stop = endsSoon || (isThin && it.ext == 0);
            visit(it.acc, it.depth, stop, it.state, stop && !endsSoon);
            if (stop) continue;
            for (int e1 = fa.edgeBegin[it.state]; e1 < fa.edgeBegin[it.state + 1]; e1++) {
                const int s1 = fa.edgeNext[e1];
                for (int e2 = fa.edgeBegin[s1]; e2 < fa.edgeBegin[s1 + 1]; e2++)
                    stack.push_back({fa.edgeNext[e2], it.depth + kLadderStep,
                                     ladderRoll(it.acc, (uint32_t)fa.edgeCh[e1] | ((uint32_t)fa.edgeCh[e2] << 8)), isThin ? it.ext - 1 : it.ext});
            }
        }
    }
};

} // namespace

static void buildFilterImpl(const Automaton &fa, Filter &f, bool allowDeep)
{
    f = Filter();
    f.shortBits.assign(65536 / 32, 0);
    const int F = fa.numPatterns;
    const int init = fa.initialState;
    auto fanout = [&](int s) { return (size_t)(fa.edgeBegin[s + 1] - fa.edgeBegin[s]); };

    size_t depth3 = 0, len3 = 0;
    bool anyShort = false;
    for (int e1 = fa.edgeBegin[init]; e1 < fa.edgeBegin[init + 1]; e1++) {
        const int s1 = fa.edgeNext[e1];
        anyShort |= s1 <= F;
        for (int e2 = fa.edgeBegin[s1]; e2 < fa.edgeBegin[s1 + 1]; e2++) {
            const int s2 = fa.edgeNext[e2];
            anyShort |= s2 <= F;
            depth3 += fanout(s2);
            for (int e3 = fa.edgeBegin[s2]; e3 < fa.edgeBegin[s2 + 1]; e3++)
                if (fa.edgeNext[e3] <= F) len3++;
        }
    }
    /* patterns below every state (a trie: children are reached from exactly one parent; post-order over an explicit stack) */
    std::vector<uint32_t> below((size_t)fa.numStates, 0u);
    if (fa.numStates > init) {
        std::vector<std::pair<int, int>> stack;                /* state, next edge */
        stack.emplace_back(init, fa.edgeBegin[init]);
        while (!stack.empty()) {
            auto &top = stack.back();
            if (top.second < fa.edgeBegin[top.first + 1]) {
                const int child = fa.edgeNext[top.second++];
                stack.emplace_back(child, fa.edgeBegin[child]);
            } else {
                const int s = top.first;
                below[s] += s <= F ? 1u : 0u;
                stack.pop_back();
                if (!stack.empty()) below[stack.back().first] += below[s];
            }
        }
    }

    /* Level 1 is tested for every position and each false positive costs list and ladder work: up to 32 KiB, two bits
     * per 3-gram.  The ladder gets up to 64 KiB.  Everything lives in LDS next to the scanning waves' queues and
     * stages (scan_filter.hip: filterLdsBytes), so the bitmaps share kFilterLdsBudget. */
    f.log2Bits = sizeLog2(depth3, 13, 18);
    f.log2BitsF3 = sizeLog2(len3, 10, 13);
    f.log2BitsLad = 19;
    auto total = [&]() {
        return kGram3LdsBytes + ((size_t(1) << f.log2BitsLad) + (size_t(1) << f.log2BitsF3)) / 8 + (anyShort ? 65536 / 8 : 0);   /* level 1 has its 32 KiB whatever its size */
    };
    while (total() > kFilterLdsBudget) {
        if (f.log2BitsLad > 17) f.log2BitsLad--;
        else if (f.log2Bits > 16) f.log2Bits--;
        else if (f.log2BitsLad > 13) f.log2BitsLad--;
        else break;
    }
    /* The ladder.  Wanted: thin = 1 (every pattern's own path is followed until it is alone on it) and one more level
     * behind a thin node; the bitmap may be a fifth full (S nodes set two bits, G nodes one or two).  If that does not
     * fit: no extra level; then a larger "thin" threshold, i.e. fewer, shallower nodes.  Then the smallest bitmap
     * that is still that sparse. */
    const int ladCap = f.log2BitsLad;
    const size_t dens = 5;
    size_t stops = 0, goOns = 0;
    auto count = [&](uint32_t thin, int ext, int last) {
        stops = goOns = 0;
        LadderWalk{fa, below, thin, ext, last}.run([&](uint32_t, int, bool stop, int, bool) { (stop ? stops : goOns)++; });
        return dens * (2 * stops + goOns) <= (size_t(1) << ladCap);
    };
    /* First choice: the ladder goes on behind kLadderLast wherever several patterns still share a path (thin nodes stop as before); that is
     * few nodes for most sets -- paths are alone long before -- and what a set with a long shared prefix needs (BASELINE config 5: 24 bytes). */
    f.ladderThin = 1;
    f.ladderExtend = 1;
    f.ladderLast = kLadderDeepLast;
#ifdef PFAC_NO_DEEP_LADDER
    const bool deep = false;
#else
    const bool deep = allowDeep && count(1, f.ladderExtend, f.ladderLast);
#endif
    if (!deep) f.ladderLast = kLadderLast;
    if (!deep && !count(1, f.ladderExtend, f.ladderLast)) {
        f.ladderExtend = 0;
        for (uint32_t thin = 1;; thin *= 2) {
            f.ladderThin = (int)thin;
            if (count(thin, 0, f.ladderLast) || thin >= (1u << 30)) break;
        }
    }
    f.log2BitsLad = 13;
    while (f.log2BitsLad < ladCap && dens * (2 * stops + goOns) > (size_t(1) << f.log2BitsLad)) f.log2BitsLad++;
    f.ladderStops = stops;
    f.ladderGoOns = goOns;

    f.gram3.assign((size_t(1) << f.log2Bits) / 32, 0);
    f.ladder.assign((size_t(1) << f.log2BitsLad) / 32, 0);
    f.final3.assign((size_t(1) << f.log2BitsF3) / 32, 0);
    auto setBit = [](std::vector<uint32_t> &v, uint32_t h) { v[h >> 5] |= 1u << (h & 31); };
    auto setGram3 = [&](uint32_t key3) { f.gram3[gram3Word(key3, f.log2Bits)] |= (1u << gram3Bit1(key3)) | (1u << gram3Bit2(key3)); };

    for (int e1 = fa.edgeBegin[init]; e1 < fa.edgeBegin[init + 1]; e1++) {
        const uint32_t c0 = fa.edgeCh[e1];
        const int s1 = fa.edgeNext[e1];
        if (s1 <= F) {                              /* 1-byte pattern */
            f.hasShort = true;
            for (uint32_t c1 = 0; c1 < 256; c1++) setBit(f.shortBits, c0 | (c1 << 8));
        }
        for (int e2 = fa.edgeBegin[s1]; e2 < fa.edgeBegin[s1 + 1]; e2++) {
            const uint32_t c1 = fa.edgeCh[e2];
            const int s2 = fa.edgeNext[e2];
            if (s2 <= F) {                          /* 2-byte pattern */
                f.hasShort = true;
                setBit(f.shortBits, c0 | (c1 << 8));
            }
            for (int e3 = fa.edgeBegin[s2]; e3 < fa.edgeBegin[s2 + 1]; e3++) {
                const uint32_t key3 = c0 | (c1 << 8) | ((uint32_t)fa.edgeCh[e3] << 16);
                setGram3(key3);
                if (fa.edgeNext[e3] <= F) {                                         /* 3-byte pattern */
                    setBit(f.final3, final3Hash(key3, f.log2BitsF3));
                    setBit(f.final3, final3Hash2(key3, f.log2BitsF3));
                }
            }
        }
    }
    /* The salt of the ladder's hashes (struct Filter).  All bits of a node share a dword now, and a GO-ON node whose two stop bits happen to be set by its
     * dword's other tenants stops every candidate that follows it -- early, at a hash no tail entry knows: a walk.  For most nodes that is one path in
     * thousands; for the node every pattern of a set with a shared prefix passes (BASELINE config 5: one path, 1 000 patterns) it is the whole stream
     * (measured on the model before the salt: depth 6 of that path was such a node, 2.5 M walks per GiB became 21.7 M).  The compiler knows every node:
     * it tries kSaltTries salts and keeps the one under which the false stops have the fewest patterns below them (0 for config 5's set). */
    f.ladderSalt = 0;
    {
        const uint32_t kSaltTries = 12;
        uint64_t bestCost = ~uint64_t(0);
        for (uint32_t t = 0; t < kSaltTries && bestCost != 0; t++) {
            const uint32_t salt = t * 0x9E3779B9u;
            std::vector<uint32_t> bits((size_t(1) << f.log2BitsLad) / 32, 0);
            std::vector<std::pair<uint32_t, uint32_t>> goOnNodes;           /* (hash, patterns below) */
            LadderWalk{fa, below, (uint32_t)f.ladderThin, f.ladderExtend, f.ladderLast, salt}.run([&](uint32_t h, int depth, bool stop, int state, bool) {
                if (stop) {
                    setBit(bits, ladderBitS1(h, f.log2BitsLad));
                    setBit(bits, ladderBitS2(h, f.log2BitsLad));
                } else {
                    setBit(bits, ladderBitG(h, f.log2BitsLad));
                    if (depth == kLadderFirst) setBit(bits, ladderBitG2(h, f.log2BitsLad));
                    goOnNodes.emplace_back(h, below[(size_t)state]);
                }
            });
            auto has = [&](uint32_t b) { return (bits[b >> 5] >> (b & 31)) & 1u; };
            uint64_t cost = 0;
            for (const auto &g : goOnNodes)
                if (has(ladderBitS1(g.first, f.log2BitsLad)) && has(ladderBitS2(g.first, f.log2BitsLad))) cost += g.second;
            if (cost < bestCost) { bestCost = cost; f.ladderSalt = salt; }
        }
    }
    struct ThinStop { uint32_t h; int depth; int state; };
    std::vector<ThinStop> thinStops;
    std::vector<uint32_t> allHashes;
    std::vector<unsigned char> goOn((size_t)fa.numStates, 0);          /* states that are G nodes of the ladder */
    std::vector<std::pair<uint32_t, int>> skipFrom;                     /* G nodes at depth kSkipFromDepth: (hash, state) */
    LadderWalk{fa, below, (uint32_t)f.ladderThin, f.ladderExtend, f.ladderLast, f.ladderSalt}.run([&](uint32_t h, int depth, bool stop, int state, bool thinStop) {
        if (!stop) {
            goOn[(size_t)state] = 1;
            if (depth == kSkipFromDepth) skipFrom.emplace_back(h, state);
        }
        if (stop) {
            setBit(f.ladder, ladderBitS1(h, f.log2BitsLad));
            setBit(f.ladder, ladderBitS2(h, f.log2BitsLad));
        } else {
            setBit(f.ladder, ladderBitG(h, f.log2BitsLad));
            if (depth == kLadderFirst) setBit(f.ladder, ladderBitG2(h, f.log2BitsLad));
        }
        allHashes.push_back(h);
        if (thinStop && below[(size_t)state] == 1) thinStops.push_back({h, depth, state});
    });
    /* skip tags (struct Filter): depth-6 G nodes with a single path of G nodes down to kLadderLast */
    f.skipCount = 0;
#ifndef PFAC_NO_SKIP_TAGS
    if (f.ladderLast >= kLadderLast) {
        std::vector<uint32_t> sortedHashes(allHashes);
        std::sort(sortedHashes.begin(), sortedHashes.end());
        for (const auto &from : skipFrom) {
            if (f.skipCount >= kSkipTagsMax) break;
            auto r = std::equal_range(sortedHashes.begin(), sortedHashes.end(), from.first);
            if (r.second - r.first != 1) continue;                      /* another ladder node has this hash */
            int s = from.second;
            bool single = true;
            for (int d = kSkipFromDepth; d < kLadderLast && single; d++) {
                /* one way on, nothing ends here; the ladder's own nodes on the way (every second depth) must be G nodes: no stop among them */
                single = s > F && fa.edgeBegin[s + 1] - fa.edgeBegin[s] == 1 && (((d - kLadderFirst) % kLadderStep) != 0 || goOn[(size_t)s]);
                if (single) s = fa.edgeNext[fa.edgeBegin[s]];
            }
            if (single) f.skipTags[f.skipCount++] = from.first;         /* (the node at kLadderLast is tested like any other) */
        }
    }
#endif
    /* the tail table (struct Filter): the rest of the one pattern below a thin stop */
    f.tail.clear();
    f.log2Tail = 0;
    f.tailEntries = 0;
#ifndef PFAC_NO_TAIL_TABLE
    if (allowDeep) {
        std::sort(allHashes.begin(), allHashes.end());
        auto shared = [&](uint32_t h) { auto r = std::equal_range(allHashes.begin(), allHashes.end(), h); return r.second - r.first > 1; };
        struct Entry { uint32_t tag, hash, info; };
        std::vector<Entry> entries;
        for (const ThinStop &t : thinStops) {
            if (shared(t.h)) continue;
            unsigned char rest[256];
            int r = 0, s2 = t.state;
            while (s2 > F && r < 255 && fa.edgeBegin[s2 + 1] - fa.edgeBegin[s2] == 1) { rest[r++] = (unsigned char)fa.edgeCh[fa.edgeBegin[s2]]; s2 = fa.edgeNext[fa.edgeBegin[s2]]; }
            if (s2 > F || fa.edgeBegin[s2 + 1] != fa.edgeBegin[s2]) continue;      /* not a single path to ONE final state without successors: leave it to the walk */
            /* a multiple of four bytes that END with the pattern (a near miss differs near the end; the rest's first bytes are left out) */
            const int all = r;
            if (all < kTailMinBytes || t.depth + all > 0xFFFF) continue;
            const int bytes = (all < kTailMaxBytes ? all : kTailMaxBytes) & ~3, skip = all - bytes;      /* four bytes a step (tailRoll) */
            uint32_t h = t.h;
            for (int i = skip; i < all; i += 4)
                h = tailRoll(h, (uint32_t)rest[i] | ((uint32_t)rest[i + 1] << 8) | ((uint32_t)rest[i + 2] << 16) | ((uint32_t)rest[i + 3] << 24));
            entries.push_back({t.h, h, (uint32_t)bytes | ((uint32_t)(t.depth + skip) << 8)});
        }
        if (!entries.empty()) {
            int lg = 8;
            while (lg < kTailLog2Max && (size_t(1) << lg) < 2 * entries.size()) lg++;
            f.log2Tail = lg;
            f.tail.assign((size_t(3) << lg), 0u);                 /* info 0 = no entry */
            for (const Entry &e : entries) {
                size_t at = (size_t)tailSlot(e.tag, lg) * 3;
                if (f.tail[at + 2] != 0) at = (size_t)tailSlot2(e.tag, lg) * 3;
                if (f.tail[at + 2] != 0) continue;
                f.tail[at] = e.tag; f.tail[at + 1] = e.hash; f.tail[at + 2] = e.info;
                f.tailEntries++;
            }
            /* the device-memory form (struct Filter: tailG): buckets of two entries, at most one entry per two buckets on average */
            int lgG = kTailGLog2Min;
            while (lgG < kTailGLog2Max && (size_t(1) << lgG) < 2 * entries.size()) lgG++;
            f.log2TailG = lgG;
            f.tailG.assign((size_t(4) << lgG), 0u);
            for (const Entry &e : entries) {
                const uint32_t bytes = e.info & 0xFFu, from = e.info >> 8;
                if (from > 255u || bytes < 4u || bytes > 32u || (bytes & 3u)) continue;
                size_t at = (size_t)tailGBucket(e.tag, lgG) * 4;
                if (f.tailG[at + 1] & kTailGFromMask) at += 2;
                if (f.tailG[at + 1] & kTailGFromMask) continue;
                f.tailG[at] = e.tag;
                f.tailG[at + 1] = (e.hash & ~kTailGInfoMask) | (from << 3) | (bytes / 4u - 1u);
                f.tailGEntries++;
            }
            f.tailCandidates = entries.size();
        }
    }
#endif
    /* Level 1 tests ONE bitmap per position: a pattern of one or two bytes matches whatever follows it, so all
     * 256 (or 65536) 3-grams that begin with it pass.  (The 2-byte bitmap is still tested at level 2, which
     * sorts out the positions this lets through.) */
    if (f.hasShort)
        for (uint32_t key2 = 0; key2 < 65536; key2++)
            if ((f.shortBits[key2 >> 5] >> (key2 & 31)) & 1u)
                for (uint32_t c2 = 0; c2 < 256; c2++) setGram3(key2 | (c2 << 16));
    f.bitsSet = f.bitsSetLad = 0;
    for (uint32_t w : f.gram3) f.bitsSet += (size_t)__builtin_popcount(w);
    for (uint32_t w : f.ladder) f.bitsSetLad += (size_t)__builtin_popcount(w);
    buildReduceFilter(fa, f);
}

void buildFilter(const Automaton &fa, Filter &f)
{
    buildFilterImpl(fa, f, /*allowDeep=*/true);
    /* The deep levels and the tail table are for the VETO kernels.  The table of a set of a few thousand patterns lies in the LDS its
     * bitmaps leave (scan_filter.hip: vetoLdsBytes; VETO = 1).  A set whose bitmaps leave none, or with more thin stops than that table
     * holds -- Snort-scale --, keeps the table in device memory (VETO = 2: one gathered load per stopped candidate, round 6); round 5 gave
     * such a set the ladder of rounds 3 and 4 and no veto at all.  A set has one form or the other. */
    const size_t lds = kGram3LdsBytes + ((size_t(1) << f.log2BitsLad) + (size_t(1) << f.log2BitsF3)) / 8 + (f.hasShort ? 65536 / 8 : 0) + f.tail.size() * sizeof(uint32_t);
    const bool inLds = !f.tail.empty() && lds <= kFilterLdsBudget && f.tailEntries * 2 >= f.tailCandidates;      /* (a table that holds less than half of the thin stops is too small: a set of tens of thousands of patterns) */
#ifdef PFAC_NO_GLOBAL_TAIL
    if ((f.ladderLast > kLadderLast || !f.tail.empty()) && !inLds) { buildFilterImpl(fa, f, /*allowDeep=*/false); return; }
#endif
    if (inLds) {
        std::vector<uint32_t>().swap(f.tailG);
        f.log2TailG = 0;
        f.tailGEntries = 0;
    } else {
        std::vector<uint32_t>().swap(f.tail);
        f.log2Tail = 0;
        f.tailEntries = 0;
        if (f.tailGEntries == 0) { std::vector<uint32_t>().swap(f.tailG); f.log2TailG = 0; }
    }
}

/* gram1 and prefix4 (struct Filter): the compacted-output kernel's level 1 -- every 3-byte prefix of a pattern, and every
 * 3-gram that begins with a pattern of one or two bytes -- and its depth-4 test -- every 4-byte prefix.  Derived from the
 * trie and shortBits alone (a compiled set on disk does not carry them). */
void buildReduceFilter(const Automaton &fa, Filter &f)
{
    f.gram1.assign((size_t(1) << kGram1Log2) / 32, 0);
    f.prefix4.assign((size_t(1) << kPrefix4Log2) / 32, 0);
    const int init = fa.initialState;
    if (fa.numStates <= init) return;
    auto setGram1 = [&](uint32_t key3) { f.gram1[gram1Word(key3)] |= 1u << gram1Bit(key3); };
    for (int e1 = fa.edgeBegin[init]; e1 < fa.edgeBegin[init + 1]; e1++) {
        const int s1 = fa.edgeNext[e1];
        for (int e2 = fa.edgeBegin[s1]; e2 < fa.edgeBegin[s1 + 1]; e2++) {
            const int s2 = fa.edgeNext[e2];
            for (int e3 = fa.edgeBegin[s2]; e3 < fa.edgeBegin[s2 + 1]; e3++) {
                const uint32_t key3 = (uint32_t)fa.edgeCh[e1] | ((uint32_t)fa.edgeCh[e2] << 8) | ((uint32_t)fa.edgeCh[e3] << 16);
                setGram1(key3);
                const int s3 = fa.edgeNext[e3];
                for (int e4 = fa.edgeBegin[s3]; e4 < fa.edgeBegin[s3 + 1]; e4++) {
                    const uint32_t h = ladderStart(key3 | ((uint32_t)fa.edgeCh[e4] << 24));       /* (prefix4 is not salted) */
                    const uint32_t b1 = prefix4Bit1(h), b2 = prefix4Bit2(h);
                    f.prefix4[b1 >> 5] |= 1u << (b1 & 31);
                    f.prefix4[b2 >> 5] |= 1u << (b2 & 31);
                }
            }
        }
    }
    if (f.hasShort && f.shortBits.size() == 65536 / 32)
        for (uint32_t key2 = 0; key2 < 65536; key2++)
            if ((f.shortBits[key2 >> 5] >> (key2 & 31)) & 1u)
                for (uint32_t c2 = 0; c2 < 256; c2++) setGram1(key2 | (c2 << 16));
}

} // namespace pfac

"""The prefilter of the scan kernel as a numpy model (test infrastructure): level 1 (blocked two-bit 3-gram bitmap), the
bypass for patterns of up to three bytes, and the prefix ladder, evaluated exactly as scan_*.hip does from the
tables the library compiled (contract: struct Filter and the hash helpers in pfac_amd/csrc/pfac_context.h)."""
import numpy as np

from pfac_amd import api

GRAM3_MUL, GRAM1_MUL, FINAL3_MUL, FINAL3_MUL2 = 0x9A17AF, 0x8B92C5, 0x85EBCB, 0xB5297B
LAD_MUL0, LAD_MUL, LAD_MULS, LAD_MULG, LAD_MULG2 = 0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D, 0x27D4EB2F, 0x165667B1
TAIL_MUL, TAIL_MUL2 = 0x9E3779B1, 0x85EBCA77
LADDER_LAST = 20                   # kLadderLast: the levels every kernel tests; a DEEP ladder (info.filterLadderLast = 60) goes on behind it


def prefilter_model(h, data, veto=True):
    """(level-1 hits, ladder candidates, positions that are walked): boolean arrays over the positions of `data`.
    veto: as the VETO kernels do it -- the levels behind LADDER_LAST of a deep ladder are tested too and a stop is put to the tail
    table; without: whatever is undecided at LADDER_LAST is walked (every other kernel)."""
    info = h.info()
    last = max(LADDER_LAST, int(getattr(info, "filterLadderLast", LADDER_LAST) or LADDER_LAST))
    g3, lad = h.table(api.PFACX_TABLE_FILTER_GRAM3), h.table(api.PFACX_TABLE_FILTER_LADDER)
    f3, sb = h.table(api.PFACX_TABLE_FILTER_FINAL3), h.table(api.PFACX_TABLE_FILTER_SHORT)
    u = np.uint64
    m32 = u(0xFFFFFFFF)
    n = data.size
    d = np.concatenate([data, np.zeros(320, dtype=np.uint8)]).astype(np.uint64)
    x = d[:n] | (d[1:n + 1] << u(8)) | (d[2:n + 2] << u(16)) | (d[3:n + 3] << u(24))

    def bit(bitmap, hv):
        return ((bitmap[(hv >> u(5)).astype(np.int64)] >> (hv & u(31)).astype(np.uint32)) & 1).astype(bool)

    # level 1: two bits of one dword (patterns of 1-2 bytes are folded into the bitmap)
    prod = ((x & u(0xFFFFFF)) * u(GRAM3_MUL)) & m32
    word = g3[((prod >> u(18)) & u((1 << (info.filterLog2Bits - 5)) - 1)).astype(np.int64)]
    level1 = (((word >> (x & u(31)).astype(np.uint32)) & (word >> ((x >> u(8)) & u(31)).astype(np.uint32))) & 1).astype(bool)
    k3 = x & u(0xFFFFFF)
    sf = u(32 - info.filterLog2BitsFinal3)
    bypass = bit(f3, ((k3 * u(FINAL3_MUL)) & m32) >> sf) & bit(f3, ((k3 * u(FINAL3_MUL2)) & m32) >> sf)
    if info.filterHasShort:
        bypass |= bit(sb, x & u(0xFFFF))
    sh = u(32 - info.filterLog2BitsLadder)

    def lad_word(hh):                      # pfac::ladderWord: all bits of a node lie in one dword (blocked, round 6)
        return lad[((hh >> u(18)) & u((1 << (info.filterLog2BitsLadder - 5)) - 1)).astype(np.int64)]

    def field(word, hh, lo):
        return ((word >> ((hh >> u(lo)) & u(31)).astype(np.uint32)) & 1).astype(bool)

    def stop(hh):
        w = lad_word(hh)
        return field(w, hh, 3) & field(w, hh, 8)

    def go_on(hh):
        return field(lad_word(hh), hh, 13)

    hh = ((x * u(LAD_MUL0)) & m32) ^ u(int(getattr(info, "filterLadderSalt", 0)))
    s, g = stop(hh), go_on(hh) & field(lad_word(hh), hh, 0)       # depth 4: G nodes set two bits
    walk = level1 & (bypass | s)
    cand = level1 & (bypass | s | g)
    und = cand & ~walk
    stop_hash = np.zeros(n, dtype=np.uint64)          # the ladder hash of the level (>= 6) at which a position was told to stop
    stopped = np.zeros(n, dtype=bool)
    skip_tags = h.table(api.PFACX_TABLE_FILTER_SKIP).astype(np.uint64)
    skipping = np.zeros(n, dtype=bool)                 # on a tagged path: not asked about the levels between 6 and LADDER_LAST
    for depth in range(6, (last if veto else LADDER_LAST) + 2, 2):
        piece = d[depth - 2:n + depth - 2] | (d[depth - 1:n + depth - 1] << u(8))
        hh = ((hh ^ piece) * u(LAD_MUL)) & m32
        if depth == 6 and skip_tags.size:
            skipping = und & np.isin(hh, skip_tags)
        asked = und & ~skipping if 6 < depth < LADDER_LAST else und
        s = stop(hh)
        first = asked & s
        stop_hash[first] = hh[first]
        stopped |= first
        walk |= first
        if depth < last:
            und = (und & ~asked) | (asked & go_on(hh) & ~s)
        else:
            und = und & False              # the ladder's last level has stops only
    walk |= und                            # a deep ladder looked at down to LADDER_LAST only: what is undecided there walks
    # the tail table: a stop whose node knows the rest of its one pattern is walked only if the candidate's bytes hash like that rest
    tail = h.table(api.PFACX_TABLE_FILTER_TAIL) if veto else np.zeros(0, dtype=np.uint32)
    if tail.size:
        lg = int(np.log2(tail.size // 3))
        t = tail.reshape(-1, 3).astype(np.uint64)
        at = np.flatnonzero(stopped)
        hs = stop_hash[at]
        slot = (((hs * u(TAIL_MUL)) & m32) >> u(32 - lg)).astype(np.int64)
        slot2 = (((hs * u(TAIL_MUL2)) & m32) >> u(32 - lg)).astype(np.int64)
        in_first = (t[slot, 2] != 0) & (t[slot, 0] == hs)
        slot = np.where(in_first, slot, slot2)
        tag, want, inf = t[slot, 0], t[slot, 1], t[slot, 2]
        has = (inf != 0) & (tag == hs)
        at, hs, want, inf = at[has], hs[has], want[has], inf[has]
        nbytes, dep = (inf & u(0xFF)).astype(np.int64), (inf >> u(8)).astype(np.int64)
        run = hs.copy()
        for i in range(0, int(nbytes.max()) if nbytes.size else 0, 4):         # pfac::tailRoll: four bytes a step
            live = i < nbytes
            p = (at + dep + i)[live]
            run[live] = ((run[live] ^ (d[p] | (d[p + 1] << u(8)) | (d[p + 2] << u(16)) | (d[p + 3] << u(24)))) * u(LAD_MUL)) & m32
        walk[at[run != want]] = False      # (the kernel also needs the bytes among those it has staged: it walks a few more)
    # ... or its device-memory form (Snort-scale sets, the VETO = 2 kernels): buckets of two 8-byte entries, 21 bits of the hash compared
    tail_g = h.table(api.PFACX_TABLE_FILTER_TAIL_GLOBAL) if veto else np.zeros(0, dtype=np.uint32)
    if tail_g.size:
        lg = int(np.log2(tail_g.size // 4))
        t = tail_g.reshape(-1, 4).astype(np.uint64)
        at = np.flatnonzero(stopped)
        hs = stop_hash[at]
        bucket = (((hs * u(TAIL_MUL)) & m32) >> u(32 - lg)).astype(np.int64)
        m1 = (t[bucket, 0] == hs) & ((t[bucket, 1] & u(0x7F8)) != 0)
        m2 = (t[bucket, 2] == hs) & ((t[bucket, 3] & u(0x7F8)) != 0)
        word = np.where(m1, t[bucket, 1], t[bucket, 3])
        has = m1 | m2
        at, hs, word = at[has], hs[has], word[has]
        nbytes, dep = (((word & u(7)) + u(1)) * u(4)).astype(np.int64), ((word >> u(3)) & u(0xFF)).astype(np.int64)
        run = hs.copy()
        for i in range(0, int(nbytes.max()) if nbytes.size else 0, 4):
            live = i < nbytes
            p = (at + dep + i)[live]
            run[live] = ((run[live] ^ (d[p] | (d[p + 1] << u(8)) | (d[p + 2] << u(16)) | (d[p + 3] << u(24)))) * u(LAD_MUL)) & m32
        walk[at[((run ^ word) & u(0xFFFFF800)) != 0] ] = False
    return level1, cand, walk


def reduce_filter_model(h, data):
    """The compacted-output kernel's filter (pfac_context.h: gram1, prefix4): (level-1 hits, positions that are walked)."""
    info = h.info()
    g1, p4 = h.table(api.PFACX_TABLE_FILTER_GRAM1), h.table(api.PFACX_TABLE_FILTER_PREFIX4)
    f3, sb = h.table(api.PFACX_TABLE_FILTER_FINAL3), h.table(api.PFACX_TABLE_FILTER_SHORT)
    assert g1.size == (1 << 19) // 32 and p4.size == (1 << 17) // 32
    u = np.uint64
    m32 = u(0xFFFFFFFF)
    n = data.size
    d = np.concatenate([data, np.zeros(8, dtype=np.uint8)]).astype(np.uint64)
    x = d[:n] | (d[1:n + 1] << u(8)) | (d[2:n + 2] << u(16)) | (d[3:n + 3] << u(24))

    def bit(bitmap, hv):
        return ((bitmap[(hv >> u(5)).astype(np.int64)] >> (hv & u(31)).astype(np.uint32)) & 1).astype(bool)

    prod = ((x & u(0xFFFFFF)) * u(GRAM1_MUL)) & m32
    word = g1[((prod >> u(18)) & u(0x3FFF)).astype(np.int64)]          # the kernel: (product's high half) & 0xFFFC as the byte address
    level1 = ((word >> (x & u(31)).astype(np.uint32)) & 1).astype(bool)
    k3 = x & u(0xFFFFFF)
    sf = u(32 - info.filterLog2BitsFinal3)
    walk = bit(f3, ((k3 * u(FINAL3_MUL)) & m32) >> sf) & bit(f3, ((k3 * u(FINAL3_MUL2)) & m32) >> sf)
    if info.filterHasShort:
        walk |= bit(sb, x & u(0xFFFF))
    hh = (x * u(LAD_MUL0)) & m32
    walk |= bit(p4, hh >> u(32 - 17)) & bit(p4, ((hh * u(LAD_MULS)) & m32) >> u(32 - 17))
    return level1, level1 & walk

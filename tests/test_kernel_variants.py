"""The kernels behind PFAC_matchFromDevice one by one: the tiled kernel in its big shape and with every bucket in LDS, the reference-layout
tables (PFACX_KERNEL_REFTABLE), long slots of wide buckets in every kernel, the two walkers of the full-result filter kernel and the
veto kernels, PFACX_WALKER_AUTO following the stream.  Reference: PFAC/src/PFAC_kernel.cu:377-458, PFAC_kernel_spaceDriven.cu:465-558."""

import concurrent.futures  # noqa: F401
import hashlib  # noqa: F401
import json  # noqa: F401
import os
import subprocess  # noqa: F401
import sys  # noqa: F401
import threading  # noqa: F401
import time  # noqa: F401

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from pfac_amd import api, sharding  # noqa: E402,F401
from pfac_amd import workloads as wl  # noqa: E402,F401
from tests.gpu_helpers import (MODES, STAGE, VARIANTS, WALKERS, assert_same, device_match, digest_record, digests, make_handle,  # noqa: E402,F401
                               o_prefix, oracle_match, perf_asserts, run_bench, timed_match)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("round4"))


@pytest.fixture(scope="module")
def mixed(workdir):
    """40 MiB of text with everything the tiled kernel branches on: sparse groups, stretches in which most positions survive
    the 3-gram test (runs of a byte that is a 1-byte pattern, a 2-byte pattern repeated), a 600-byte and a 2000-byte pattern
    planted across the 4 KiB group boundaries and more than 128 bytes (the staged halo) deep, complete and with a wrong last
    byte, and patterns that end exactly at and just beyond the last byte."""
    from oracle import binding as ob
    rng = np.random.Generator(np.random.PCG64(404))
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789 /.-_=&%:", dtype=np.uint8)
    pats = {b"q", b"zz", b"%%"}
    while len(pats) < 1500:
        pats.add(alpha[rng.integers(0, alpha.size, int(rng.integers(3, 40)))].tobytes())
    p600 = alpha[rng.integers(0, alpha.size, 600)].tobytes()
    p2000 = alpha[rng.integers(0, alpha.size, 2000)].tobytes()
    pats = sorted(pats) + [p600, p2000, p600[:150] + b"#"]
    pf = wl.write_pattern_file(os.path.join(workdir, "mixed.pat"), pats)
    n = (40 << 20) + 1237
    data = alpha[rng.integers(0, alpha.size, n)].copy()
    for k in range(24):                                          # dense stretches: every position matches 'q', or 'zz' at every position
        at = int(rng.integers(0, n - (1 << 16)))
        data[at:at + (8192 if k % 2 else 40000)] = ord("q") if k % 3 else ord("z")
    for k, at in enumerate([4096 - 300, 8192 - 1999, (1 << 20) - 64, (9 << 20) - 130, (17 << 20) + 4096 - 599, n - 2000, n - 600, n - 2001]):
        p = np.frombuffer(p2000 if k % 2 else p600, dtype=np.uint8)
        at = min(at, n - p.size)
        data[at:at + p.size] = p
        if k % 3 == 2:
            data[at + p.size - 1] = ord("#")                      # near miss: the whole pattern is walked, a shorter one (or none) is reported
    data[n - 1] = ord("q")                                        # a 1-byte pattern on the very last byte
    data[n - 3:n - 1] = np.frombuffer(b"zz", dtype=np.uint8)
    o = ob.Oracle(pf)
    want = o.match(data, omp=True)
    o.close()
    assert np.count_nonzero(want) > n // 200 and want[n - 1] != 0
    return pf, data, want


@pytest.mark.parametrize("perf,tex,mode_name", MODES)
def test_tiled_kernel_big_shape_equals_oracle(mixed, perf, tex, mode_name):
    """PFACX_KERNEL_NAIVE on 40 MiB: the tiled kernel's big shape (1024-thread blocks, 4 KiB groups, hot rows in LDS), aligned
    and misaligned pointers (groups are cut at 16-byte addresses: masked positions in front of the first byte and behind the
    last), sparse and dense groups, walks beyond the staged halo, the end of the input."""
    pf, data, want = mixed
    h = make_handle(pf, perf, tex, api.PFACX_KERNEL_NAIVE)
    try:
        assert_same(device_match(h, data), want, f"tiled / {mode_name} / aligned")
        assert_same(device_match(h, data, in_offset=5, out_offset=3), want, f"tiled / {mode_name} / input +5 B, result +3 ints")
        # the default variant at a size it gives to the tiled kernel (below 32 MiB) and at one it gives to the filter kernel,
        # whose dense chunks come back to the tiled kernel
        h.setKernelVariant(api.PFACX_KERNEL_AUTO)
        m = 20 << 20
        assert_same(device_match(h, data[:m + 2500])[:m], want[:m], f"auto, 20 MiB / {mode_name}")
        assert_same(device_match(h, data), want, f"auto, 40 MiB / {mode_name}")
        st = h.scanStats(data.size)
        assert st["denseChunks"] > 0, st                          # the 'q' and 'z' stretches
    finally:
        h.destroy()


def test_tiled_kernel_compacted_output_equals_the_full_vector(mixed):
    """PFAC_matchFromDeviceReduce through the tiled kernel (PFACX_KERNEL_NAIVE; what AUTO does below 32 MiB): the pairs are the
    non-zero results of the full vector, in position order, for the big and for the small shape."""
    pf, data, want = mixed
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_ON, api.PFACX_KERNEL_NAIVE)
    try:
        for n in (data.size, (3 << 20) + 17):
            part = data[:n]
            d_in = torch.from_numpy(part.copy()).to("cuda:0")
            d_res = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
            d_pos = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
            _, count = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_res.data_ptr(), d_pos.data_ptr())
            # the oracle's vector, restricted to matches that fit into the first n bytes: results near the cut may be shorter patterns
            from oracle import binding as ob
            o = ob.Oracle(pf)
            ref = o.match(part, omp=True)
            o.close()
            pos = np.flatnonzero(ref)
            assert count == pos.size
            assert np.array_equal(d_pos[:count].cpu().numpy(), pos) and np.array_equal(d_res[:count].cpu().numpy(), ref[pos])
    finally:
        h.destroy()


def test_reference_layout_tables_exist_on_the_device_only_on_request(workdir):
    """BASELINE config 3's pattern set under the DEFAULT perf mode (PFAC_TIME_DRIVEN): the dense table of the reference would be
    498 MB on the host and on the device; no product kernel reads it, so it is not built -- the set holds < 32 MB on the
    device.  PFACX_KERNEL_REFTABLE builds and uploads it (and gives the same results); PFAC_setPerfMode semantics
    (PFAC.cpp:794-814: the tables follow the mode) are unchanged."""
    cfg = wl.make_config("c3")
    pf = wl.write_pattern_file(os.path.join(workdir, "c3.pat"), cfg.patterns)
    data = cfg.input_slice(2 << 20, 0)
    h = api.PFAC.create()
    try:
        h.readPatternFromFile(pf)                                 # defaults: TIME_DRIVEN, AUTOMATIC, AUTO
        info = h.info()
        assert info.perfMode == api.PFAC_TIME_DRIVEN and info.sizeOfTableInBytes == 256 * 4 * info.numOfStates > 400e6
        assert 0 < info.deviceTableBytes < 32e6, info.deviceTableBytes
        base = device_match(h, data)
        h.setKernelVariant(api.PFACX_KERNEL_REFTABLE)             # the reference-shaped kernel walks int[S][256]: now it exists
        assert h.info().deviceTableBytes > 400e6
        assert_same(device_match(h, data), base, "reftable, dense")
        h.setPerfMode(api.PFAC_SPACE_DRIVEN)                      # the tables follow the mode: hashed pair, a few MB
        assert h.info().deviceTableBytes < 48e6 and h.info().sizeOfTableEntry == 8
        assert_same(device_match(h, data), base, "reftable, hashed")
        h.setKernelVariant(api.PFACX_KERNEL_AUTO)
        h.setPerfMode(api.PFAC_TIME_DRIVEN)
        assert h.info().deviceTableBytes < 32e6
        assert_same(device_match(h, data), base, "auto again")
        assert h.table(api.PFACX_TABLE_DENSE).size == 256 * info.numOfStates     # the host copy: built on first use
    finally:
        h.destroy()


@pytest.mark.parametrize("perf,tex,mode_name", [MODES[1], MODES[2]])
def test_tiled_kernel_with_every_bucket_in_lds(workdir, perf, tex, mode_name):
    """A pattern set whose whole chained table fits the CU's LDS takes the HOTALL instance of the tiled kernel's big shape (no
    global path in a walk step): the README's four patterns plus a few longer ones over 24 MiB of text in which they occur
    sparsely, and over a stretch in which they occur at every position (dense groups), full and compacted output."""
    from oracle import binding as ob
    rng = np.random.Generator(np.random.PCG64(77))
    pats = [b"AB", b"ABG", b"BEDE", b"ED", b"GATTACA", b"EDEDEDEDEDEDEDEDED", b"ABGABGAB"]
    pf = wl.write_pattern_file(os.path.join(workdir, "tiny.pat"), pats)
    n = (24 << 20) + 333
    alpha = np.frombuffer(b"ABCDEGT", dtype=np.uint8)
    data = np.frombuffer(b"xyzw", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
    for at in rng.integers(0, n - 64, 20000):                      # sparse occurrences
        k = int(rng.integers(4, 40))
        data[at:at + k] = alpha[rng.integers(0, alpha.size, k)]
    data[5 << 20:(5 << 20) + 300000] = np.frombuffer(b"ED", dtype=np.uint8)[np.arange(300000) % 2]   # dense: a match at every position
    o = ob.Oracle(pf)
    want = o.match(data, omp=True)
    o.close()
    h = make_handle(pf, perf, tex, api.PFACX_KERNEL_NAIVE)
    try:
        assert_same(device_match(h, data), want, f"hotall / {mode_name}")
        assert_same(device_match(h, data, in_offset=9, out_offset=1), want, f"hotall / {mode_name} / misaligned")
        d_in = torch.from_numpy(data).to("cuda:0")
        d_res = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
        d_pos = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
        _, count = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_res.data_ptr(), d_pos.data_ptr())
        pos = np.flatnonzero(want)
        assert count == pos.size and np.array_equal(d_pos[:count].cpu().numpy(), pos) and np.array_equal(d_res[:count].cpu().numpy(), want[pos])
    finally:
        h.destroy()


@pytest.fixture(scope="module")
def longset(workdir):
    """Patterns whose tries are long single-successor runs cut in every way a slot can be cut: lengths 9 .. 64 in steps of one
    (chains of every length 0 .. 23 behind a branch), one 300-byte and one 700-byte pattern (several long slots in a row; deeper
    than the 128 bytes staged behind a chunk), patterns that are prefixes of patterns at depths 8, 9, 24, 25 (a final state with
    successors ends a slot early), a shared 24-byte prefix with 40 tails (BASELINE config 5's shape) and a few short ones."""
    rng = np.random.Generator(np.random.PCG64(55))
    low = np.arange(97, 123, dtype=np.uint8)
    def word(n):
        return low[rng.integers(0, low.size, n)].tobytes()
    pats = set()
    for n in range(9, 65):
        pats.add(word(n))
    p300, p700 = word(300), word(700)
    pats.update([p300, p700, p300[:8], p300[:9], p300[:24], p300[:25], p700[:100], p700[:101] + b"X"])
    prefix = word(24)
    tails = [word(int(rng.integers(8, 41))) for _ in range(40)]
    pats.update(prefix + t for t in tails)
    pats.update([b"zq", b"q", b"zqzqzq"])
    pats = sorted(pats)
    pf = wl.write_pattern_file(os.path.join(workdir, "longset.pat"), pats)
    return pf, pats, prefix, tails, p300, p700


def _plant_stream(pats, prefix, tails, p300, p700, n, seed, density):
    """filler that matches nothing but 'q' now and then, with complete patterns, near misses (last 1..4 bytes wrong) and truncated
    patterns planted every `density` bytes on average, at offsets that sweep the chunk / tile / lane boundaries"""
    rng = np.random.Generator(np.random.PCG64(seed))
    data = (rng.integers(0, 6, n, dtype=np.uint8) + 48).astype(np.uint8)           # '0'..'5'
    data[rng.integers(0, n, n // 997)] = ord("q")
    pool = [p for p in pats if len(p) >= 9] + [prefix + t for t in tails] * 3 + [p300, p700]
    at = 7
    k = 0
    while at + 800 < n:
        p = pool[int(rng.integers(0, len(pool)))]
        kind = k % 4
        if kind == 1:
            cut = int(rng.integers(1, 5))
            p = p[:-cut] + b"#" * cut                                               # near miss: walked almost to the end
        elif kind == 2:
            p = p[:int(rng.integers(1, len(p)))]                                    # truncated: the input goes on with filler
        data[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
        at += len(p) + int(rng.integers(0, 2 * density))
        k += 1
    return data


@pytest.mark.parametrize("perf,tex,mode_name", MODES)
@pytest.mark.parametrize("variant,variant_name", [(api.PFACX_KERNEL_FILTER | (api.PFACX_WALKER_WINDOW << 8), "filter-window"),
                                                  (api.PFACX_KERNEL_FILTER | (api.PFACX_WALKER_STAGE << 8), "filter-stage"),
                                                  (api.PFACX_KERNEL_FILTER | (api.PFACX_WALKER_VETO << 8), "filter-veto"),
                                                  (api.PFACX_KERNEL_NAIVE, "tiled"),                                        # the narrow table: no slot is long
                                                  (api.PFACX_KERNEL_NAIVE | (api.PFACX_WALKER_STAGE << 8), "tiled-wide"),  # a caller that expects near misses: the wide table, units on demand
                                                  (api.PFACX_KERNEL_REFTABLE, "reftable")])
def test_long_slots_in_every_kernel(workdir, longset, perf, tex, mode_name, variant, variant_name):
    pf, pats, prefix, tails, p300, p700 = longset
    n = 2048 * 150 + 333
    data = _plant_stream(pats, prefix, tails, p300, p700, n, seed=1, density=40)
    # patterns across every 2 KiB chunk boundary at every distance 0 .. 99, and up against the end of the input
    for j in range(100):
        at = 2048 * (10 + j) - j
        p = (prefix + tails[j % len(tails)]) if j % 3 else (p300 if j % 2 else p700[:200])
        data[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
    data[n - 300:] = np.frombuffer(p300, dtype=np.uint8)
    want = oracle_match(pf, data)
    assert np.count_nonzero(want) > 2000
    h = make_handle(pf, perf, tex, variant)
    try:
        assert_same(device_match(h, data), want, f"long slots/{mode_name}/{variant_name}")
        # compacted output walks the same table (register window; extension units fetched on demand)
        d_in = torch.from_numpy(data).to("cuda:0")
        d_ids = torch.full((n,), -3, dtype=torch.int32, device="cuda:0")
        d_pos = torch.full((n,), -3, dtype=torch.int32, device="cuda:0")
        _, count = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_ids.data_ptr(), d_pos.data_ptr())
        nz = np.nonzero(want)[0]
        assert count == nz.size
        assert np.array_equal(d_pos[:count].cpu().numpy(), nz) and np.array_equal(d_ids[:count].cpu().numpy(), want[nz])
    finally:
        h.destroy()


@pytest.fixture(scope="module")
def switching(workdir, longset):
    """48 MiB in which stretches of plain filler (a few MiB: text mode) alternate with stretches full of near misses of long
    patterns (a walk every ~60 bytes, 30..60 bytes deep: stage mode), so that every scanning wave changes its mode several
    times inside ONE launch of the filter kernel; patterns straddle the places where the stream changes its nature."""
    pf, pats, prefix, tails, p300, p700 = longset
    n = (48 << 20) + 4099
    parts = []
    seed = 100
    left = n
    while left > 0:
        for density, size in ((4000, 5 << 20), (12, 3 << 20), (100000, 2 << 20), (6, 1 << 20)):
            size = min(size, left)
            if size <= 0:
                break
            parts.append(_plant_stream(pats, prefix, tails, p300, p700, size, seed, density) if size > 2000 else np.full(size, 48, np.uint8))
            seed += 1
            left -= size
    data = np.concatenate(parts)
    assert data.size == n
    want = oracle_match(pf, data, omp=True)
    return pf, data, want


@pytest.mark.parametrize("walker,walker_name", [(api.PFACX_WALKER_AUTO, "auto")] + WALKERS)
def test_stream_that_changes_its_nature_inside_one_launch(switching, walker, walker_name):
    pf, data, want = switching
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_ON, api.PFACX_KERNEL_AUTO | (walker << 8))
    h.setWalker(walker)                            # AUTO too: a session under PFAC_TEST_WALKER has every new handle's walker forced
    try:
        for call in range(3):                      # AUTO: the first call runs the window walker, the next ones what the votes say
            assert_same(device_match(h, data), want, f"changing stream/{walker_name}/call {call}")
            st = h.scanStats()
            if walker == api.PFACX_WALKER_STAGE:
                assert st["walker"] == api.PFACX_WALKER_STAGE
            if walker == api.PFACX_WALKER_WINDOW or (walker == api.PFACX_WALKER_AUTO and call == 0):
                assert st["walker"] == api.PFACX_WALKER_WINDOW
    finally:
        h.destroy()


@pytest.mark.parametrize("extra,vetoes", [(20000, False), (2000, True)])
def test_auto_walker_follows_the_stream(workdir, extra, vetoes):
    """PFACX_WALKER_AUTO.  A pattern set too large for a tail table in LDS (pfac_context.h: the Snort-scale filter bitmaps take it) keeps
    the table in device memory: a handle's first full-result launch runs the plain register-window walker; after a launch over a stream
    full of near misses (most scanning waves end it having met long slots) the next one runs the veto kernel (VETO = 2: a stop of the
    ladder is put to the table with one gathered load), which goes on reporting near misses -- the candidates it vetoes -- while the
    stream stays so; after a launch over text the plain window walker again.  A set of a few thousand patterns has the table in LDS:
    every launch runs the window walker behind that veto (VETO = 1).  Either way near misses hardly reach a walker.  Results are the
    oracle's throughout (here: the committed small C5 / C3 generators, 64 MiB each, against the oracle)."""
    cfg5, cfg3 = wl.make_config("c5"), wl.make_config("c3")
    # one pattern set for both streams: the near-miss patterns + a slice of the Snort-style set
    pats = list(cfg5.patterns) + [p for p in cfg3.patterns[:extra] if p not in set(cfg5.patterns)]
    pf = wl.write_pattern_file(os.path.join(workdir, f"auto{extra}.pat"), pats)
    n = 64 << 20
    near, text = cfg5.input_slice(n, 0), cfg3.input_slice(n, 0)
    want_near, want_text = oracle_match(pf, near, omp=True), oracle_match(pf, text, omp=True)
    h = make_handle(pf, api.PFAC_TIME_DRIVEN, api.PFAC_TEXTURE_ON, api.PFACX_KERNEL_AUTO)
    h.setWalker(api.PFACX_WALKER_AUTO)             # (a session under PFAC_TEST_WALKER forces every new handle's walker)
    try:
        info = h.info()
        assert (info.filterTailEntries > 0) == vetoes and (info.filterTailGlobalEntries > 0) == (not vetoes), (info.filterTailEntries, info.filterTailGlobalEntries)
        seen = []
        for stream, want, name in ((text, want_text, "text"), (near, want_near, "near"), (near, want_near, "near"), (near, want_near, "near"),
                                   (text, want_text, "text"), (text, want_text, "text"), (text, want_text, "text")):
            assert_same(device_match(h, stream), want, f"auto walker/{name}")
            st = h.scanStats()
            seen.append((name, st["walker"], st["stageModeWaves"], st["walksStarted"], st["veto"]))
        walkers = [w for _, w, _, _, _ in seen]
        vetoed = [v for _, _, _, _, v in seen]
        W = api.PFACX_WALKER_WINDOW
        assert walkers == [W] * 7, seen                           # the stage walker is what a set WITHOUT a tail table of either form gets (and PFACX_WALKER_STAGE)
        if vetoes:
            assert vetoed == [1] * 7, seen
            assert seen[2][3] * 3 < (n >> 20) * 22000, seen      # the near-miss stream: a third of the 22 K candidates per MiB walk at most
        else:
            assert vetoed[0] == 0 and vetoed[1] == 0, seen        # text first; the near-miss stream's first launch still has the text verdict
            assert vetoed[2] == 2 and vetoed[3] == 2, seen        # ... its next launches run the veto kernel, which keeps the verdict up
            assert vetoed[4] == 2, seen                           # the first text launch behind it: still the near-miss verdict
            assert vetoed[5] == 0 and vetoed[6] == 0, seen        # and back
            assert seen[1][2] > 0 and seen[3][2] > 0 and seen[6][2] == 0, seen
            assert seen[1][3] > (n >> 20) * 15000 and seen[3][3] * 3 < (n >> 20) * 22000, seen      # walks: every near miss without the veto, a third at most behind it
    finally:
        h.destroy()


@pytest.fixture(scope="module")
def bigset(workdir):
    """A Snort-scale set (C3's 30 000 patterns + 300 patterns that share a 24-byte prefix + a 300-byte pattern) whose tail table lies in
    device memory, and 48 MiB + 13 of a stream that is text in its first third, near misses of the shared-prefix patterns in the second
    (last 1..4 bytes wrong, complete ones in between, some cut by the 2 KiB chunk boundaries and more than the staged 48 bytes beyond
    them) and a mix in the third."""
    rng = np.random.Generator(np.random.PCG64(606))
    alnum = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789", dtype=np.uint8)
    cfg3 = wl.make_config("c3")
    prefix = alnum[rng.integers(0, alnum.size, 24)].tobytes()
    shared = sorted({prefix + alnum[rng.integers(0, alnum.size, int(rng.integers(8, 41)))].tobytes() for _ in range(300)})
    p300 = alnum[rng.integers(0, alnum.size, 300)].tobytes()
    seen = set(cfg3.patterns)
    pats = list(cfg3.patterns) + [p for p in shared + [p300] if p not in seen]
    pf = wl.write_pattern_file(os.path.join(workdir, "bigset.pat"), pats)
    third = 16 << 20
    n = 3 * third + 13
    data = np.empty(n, dtype=np.uint8)
    data[:third] = cfg3.input_slice(third, 0)
    recs = []
    for k in range(4096):
        p = shared[int(rng.integers(0, len(shared)))] if k % 97 else p300
        miss = int(rng.integers(0, 5)) if k % 5 else 0                 # one record in five is a complete pattern
        recs.append(np.frombuffer(p[:len(p) - miss], dtype=np.uint8))
        recs.append(alnum[rng.integers(0, alnum.size, miss + int(rng.integers(0, 3)))])
    pool = np.concatenate(recs)
    reps = -(-2 * third // pool.size)
    near = np.tile(pool, reps)[:2 * third + 13].copy()
    data[third:] = near
    mix = data[2 * third:]
    for k in range(200):                                                # text islands in the last third
        at = int(rng.integers(0, mix.size - 70000))
        mix[at:at + 65536] = cfg3.input_slice(65536, 1 + k % 3)
    # 2 MiB of the SHORTEST records back to back (32 bytes: 64 near misses per 2 KiB chunk, plus whatever the other 30 000 patterns' prefixes add):
    # more candidates than one ladder batch takes, so a second batch of the same trip runs while the first one's answers are on their way and has
    # to leave them their queue entries
    short = sorted(shared, key=len)[:40]
    srecs = []
    for k in range(70000):
        p = short[int(rng.integers(0, len(short)))]
        miss = int(rng.integers(1, 4)) if k % 7 else 0
        srecs.append(np.frombuffer(p[:len(p) - miss], dtype=np.uint8))
        srecs.append(alnum[rng.integers(0, alnum.size, miss)])
    sblock = np.concatenate(srecs)[:2 << 20]
    data[n - (5 << 20):n - (5 << 20) + sblock.size] = sblock
    # three blocks of pattern PREFIXES back to back inside the text third: 5 bytes of a pattern + 6..9 / 2..4 / 0..1 other bytes -- ~170 / 257..307 / 310..388
    # level-1 hits per 2 KiB chunk (tests/filter_model.py), 200..320 of them candidates behind the level-4 test: around and above the list's codes (128; the
    # VETO = 2 instance: 256), so a chunk takes several list rounds with leftover candidates carried from one to the next and several ladder batches per trip --
    # and below the 1024 hits at which a chunk goes to the tiled kernel instead
    longer = [q for q in cfg3.patterns if len(q) >= 8]
    for blk, (gap_lo, gap_hi) in enumerate(((6, 10), (2, 5), (0, 2))):
        precs = []
        for k in range(120000):
            q = longer[int(rng.integers(0, len(longer)))]
            precs.append(np.frombuffer(q[:5] if k % 11 else q, dtype=np.uint8))        # one in eleven: the whole pattern
            precs.append(alnum[rng.integers(0, alnum.size, int(rng.integers(gap_lo, gap_hi)))])
        pblock = np.concatenate(precs)[:512 << 10]
        at = (3 + 2 * blk) << 20
        data[at:at + pblock.size] = pblock
    for k in range(64):                                                 # complete patterns across chunk boundaries, ending 49..63 bytes beyond them
        p = np.frombuffer(shared[k % len(shared)], dtype=np.uint8)
        at = third + 2048 * (100 + 37 * k) - (p.size - 49 - k % 15)
        data[at:at + p.size] = p
    want = oracle_match(pf, data, omp=True)
    assert np.count_nonzero(want) > 50000
    return pf, data, want, pats


@pytest.mark.parametrize("perf,tex,mode_name", MODES)
def test_veto_kernel_with_the_tail_table_in_device_memory(bigset, perf, tex, mode_name):
    """The VETO = 2 instances (PFACX_WALKER_VETO on a set whose tail table lies in device memory) against the oracle in every table mode:
    aligned and misaligned pointers, and the same handle under PFACX_WALKER_AUTO across a stream that changes from text to near misses
    and back inside ONE launch (the kernel's per-batch gate: a batch with fewer than eight stopped candidates does not ask the table)."""
    pf, data, want, _ = bigset
    h = make_handle(pf, perf, tex, api.PFACX_KERNEL_FILTER | (api.PFACX_WALKER_VETO << 8))
    try:
        info = h.info()
        assert info.filterTailGlobalEntries > 10000 and info.filterTailEntries == 0 and info.filterLadderLast > 20, (info.filterTailGlobalEntries, info.filterLadderLast)
        assert_same(device_match(h, data), want, f"veto kernel / {mode_name} / aligned")
        st = h.scanStats(data.size)
        assert st["veto"] == 2 and st["walksStarted"] > 0, st
        assert_same(device_match(h, data, in_offset=3, out_offset=1), want, f"veto kernel / {mode_name} / input +3 B, result +1 int")
        plain = make_handle(pf, perf, tex, api.PFACX_KERNEL_FILTER | (api.PFACX_WALKER_WINDOW << 8))
        try:
            assert_same(device_match(plain, data), want, f"window walker, same set / {mode_name}")
            st0 = plain.scanStats(data.size)
            assert st0["veto"] == 0 and st["walksStarted"] * 3 < st0["walksStarted"] * 2, (st["walksStarted"], st0["walksStarted"])      # the veto spares the walks of the near misses (a fifth of the records are complete patterns, text walks as before)
        finally:
            plain.destroy()
        h.setWalker(api.PFACX_WALKER_AUTO)
        for k in range(3):
            assert_same(device_match(h, data), want, f"auto / {mode_name} / launch {k}")
    finally:
        h.destroy()


def test_veto_kernel_with_the_tail_table_in_device_memory_and_short_patterns(bigset, workdir):
    """The HAS_SHORT instances of the VETO = 2 kernel: the same Snort-scale set plus a one-byte and a two-byte pattern (the exact 2-byte bitmap
    in LDS, the bypass in the level-4 test), over 2 MiB each of the fixture's prefix blocks, near misses and short records with the two
    short patterns planted in them -- against the oracle, aligned and misaligned."""
    _, big, _, pats = bigset
    pf = wl.write_pattern_file(os.path.join(workdir, "bigset_short.pat"), list(pats) + [b"\x01\x02", b"\x7f"])
    third = 16 << 20
    n = big.size
    data = np.concatenate([big[(3 << 20):(5 << 20)], big[third + (1 << 20):third + (3 << 20)], big[n - (5 << 20):n - (3 << 20)], big[:13]]).copy()
    rng = np.random.Generator(np.random.PCG64(607))
    for at in rng.integers(0, data.size - 2, 3000):
        if at % 3:
            data[at] = 0x7F
        else:
            data[at:at + 2] = (1, 2)
    want = oracle_match(pf, data, omp=True)
    short_hits = int(np.count_nonzero(data == 0x7F))
    assert np.count_nonzero(want) > short_hits > 1500
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_ON, api.PFACX_KERNEL_FILTER | (api.PFACX_WALKER_VETO << 8))
    try:
        info = h.info()
        assert info.filterHasShort and info.filterTailGlobalEntries > 10000 and info.filterTailEntries == 0, (info.filterHasShort, info.filterTailGlobalEntries)
        assert_same(device_match(h, data), want, "veto kernel, short patterns / aligned")
        st = h.scanStats(data.size)
        assert st["veto"] == 2 and st["walksStarted"] > 0, st
        assert_same(device_match(h, data, in_offset=5, out_offset=3), want, "veto kernel, short patterns / input +5 B, result +3 ints")
    finally:
        h.destroy()

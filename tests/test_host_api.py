"""Host side of the C ABI, no GPU needed.

Covers (a) the exported surface: every symbol include/*.h declares is present in libpfac.so /
libpfac_gfx950.so; (b) the reference's status-code behaviour (PFAC/src/PFAC.cpp argument checks);
(c) the pattern compiler and both table materialisers byte-for-byte against the oracle;
(d) the CPU platforms (PFAC_PLATFORM_CPU / CPU_OMP) against the oracle; (e) the prefilter bitmaps
are supersets (a miss proves the result is 0).  Everything goes through a PFACX_createHostOnly
handle: no compute call touches a device.
"""

import ctypes
import os
import re

import numpy as np
import pytest

from oracle import binding as ob
from pfac_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    lib = api.load_library()
    host, module = api.library_paths()
    declared_host, declared_module = set(), set()
    for header, bucket in (("PFAC.h", declared_host), ("pfac_ext.h", declared_host), ("pfac_module.h", declared_module)):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for name in re.findall(r"\b(PFACX?_[A-Za-z_]+)\s*\(", text):
            if not name.endswith("protoType") and not name.endswith("_t") and not name.isupper():
                bucket.add(name)
    assert declared_host == set(api.EXPORTED_SYMBOLS), declared_host ^ set(api.EXPORTED_SYMBOLS)
    assert declared_module == set(api.MODULE_SYMBOLS), declared_module ^ set(api.MODULE_SYMBOLS)
    for name in declared_host:
        assert getattr(lib, name)
    mod = ctypes.CDLL(module)
    for name in declared_module:
        assert getattr(mod, name)


def test_status_values_and_error_strings():
    """Enum values of PFAC.h:57-70 and the texts of PFAC.cpp:1131-1183."""
    S = api.STATUS
    assert (S.SUCCESS, S.BASE, S.ALLOC_FAILED, S.CUDA_ALLOC_FAILED, S.INVALID_HANDLE, S.INVALID_PARAMETER,
            S.PATTERNS_NOT_READY, S.FILE_OPEN_ERROR, S.LIB_NOT_EXIST, S.ARCH_MISMATCH, S.MUTEX_ERROR,
            S.INTERNAL_ERROR) == (0, 10000, 10001, 10002, 10003, 10004, 10005, 10006, 10007, 10008, 10009, 10010)
    assert api.error_string(S.SUCCESS).startswith("PFAC_STATUS_SUCCESS")
    assert api.error_string(S.PATTERNS_NOT_READY).startswith("PFAC_STATUS_PATTERNS_NOT_READY")
    assert api.error_string(S.FILE_OPEN_ERROR) == "PFAC_STATUS_FILE_OPEN_ERROR: pattern file does not exist"
    assert api.error_string(10999).startswith("PFAC_STATUS_INTERNAL_ERROR")      # default branch


def test_null_handle_and_argument_checks(workloads, tmp_path):
    lib = api.load_library()
    null = ctypes.c_void_p()
    assert lib.PFAC_destroy(null) == api.STATUS.INVALID_HANDLE
    assert lib.PFAC_setPlatform(null, 0) == api.STATUS.INVALID_HANDLE
    assert lib.PFAC_setPerfMode(null, 0) == api.STATUS.INVALID_HANDLE
    assert lib.PFAC_setTextureMode(null, 0) == api.STATUS.INVALID_HANDLE
    assert lib.PFAC_readPatternFromFile(null, b"x") == api.STATUS.INVALID_HANDLE
    assert lib.PFAC_matchFromHost(null, 1, 1, 1) == api.STATUS.INVALID_HANDLE
    assert lib.PFAC_matchFromDevice(null, 1, 1, 1) == api.STATUS.INVALID_HANDLE
    assert lib.PFAC_dumpTransitionTable(null, None) == api.STATUS.INVALID_HANDLE

    h = api.PFAC.createHostOnly()
    assert h.setPlatform(7, check=False) == api.STATUS.INVALID_PARAMETER
    assert h.setPerfMode(2, check=False) == api.STATUS.INVALID_PARAMETER
    assert h.setTextureMode(3, check=False) == api.STATUS.INVALID_PARAMETER
    buf = np.zeros(16, dtype=np.int32)
    # order of checks: ready, input, output, size (ref PFAC.cpp:882-897)
    assert h.matchFromHost(buf.ctypes.data, 4, buf.ctypes.data, check=False) == api.STATUS.PATTERNS_NOT_READY
    assert h.readPatternFromFile(None, check=False) == api.STATUS.INVALID_PARAMETER
    assert h.readPatternFromFile(str(tmp_path / "missing.pat"), check=False) == api.STATUS.FILE_OPEN_ERROR
    assert h.readPatternFromFile("x" * 300, check=False) == api.STATUS.INTERNAL_ERROR     # FILENAME_LEN, PFAC.cpp:668-672
    h.readPatternFromFile(workloads["c1"].pattern_file)
    assert h.matchFromHost(0, 4, buf.ctypes.data, check=False) == api.STATUS.INVALID_PARAMETER
    assert h.matchFromHost(buf.ctypes.data, 4, 0, check=False) == api.STATUS.INVALID_PARAMETER
    buf[:] = -3
    assert h.matchFromHost(buf.ctypes.data, 0, buf.ctypes.data, check=False) == api.STATUS.SUCCESS
    assert np.all(buf == -3), "size 0 must not write"
    # a host-only handle has no GPU path and must say so loudly (no silent CPU fallback)
    assert h.matchFromDevice(buf.ctypes.data, 4, buf.ctypes.data, check=False) == api.STATUS.LIB_NOT_EXIST
    h.setPlatform(api.PFAC_PLATFORM_GPU)
    assert h.matchFromHost(buf.ctypes.data, 4, buf.ctypes.data, check=False) == api.STATUS.LIB_NOT_EXIST
    # pfac_ext.h: the walker choice and the multi-GPU calls check like the rest (no device here: LIB_NOT_EXIST behind the argument checks)
    assert lib.PFACX_setWalker(null, api.PFACX_WALKER_AUTO) == api.STATUS.INVALID_HANDLE
    assert h.setWalker(5, check=False) == api.STATUS.INVALID_PARAMETER
    for w in (api.PFACX_WALKER_WINDOW, api.PFACX_WALKER_STAGE, api.PFACX_WALKER_AUTO):
        assert h.setWalker(w, check=False) == api.STATUS.SUCCESS
    cnt = ctypes.c_int(-1)
    assert lib.PFACX_matchFromHostReduceMultiGPU(null, 1, 1, 1, 1, ctypes.byref(cnt), 0, None) == api.STATUS.INVALID_HANDLE
    st, count = h.matchFromHostReduceMultiGPU(0, 4, buf.ctypes.data, buf.ctypes.data, check=False)
    assert st == api.STATUS.INVALID_PARAMETER
    st, count = h.matchFromHostReduceMultiGPU(buf.ctypes.data, 0, buf.ctypes.data, buf.ctypes.data, check=False)
    assert st == api.STATUS.SUCCESS and count == 0
    st, count = h.matchFromHostReduceMultiGPU(buf.ctypes.data, 4, buf.ctypes.data, buf.ctypes.data, [0], check=False)
    assert st == api.STATUS.LIB_NOT_EXIST
    assert h.matchFromHostMultiGPU(buf.ctypes.data, 4, buf.ctypes.data, [0], check=False) == api.STATUS.LIB_NOT_EXIST
    assert h.destroy() == api.STATUS.SUCCESS


def test_create_without_device_forwards_the_runtime_error():
    """ref PFAC.cpp:148-151: PFAC_create returns the raw runtime error code when no device is usable."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    h = api.PFAC.create(check=False)
    assert 0 < h.create_status < api.STATUS.BASE
    assert api.error_string(h.create_status)


@pytest.mark.parametrize("name", ["c1", "ex2", "c2", "c3", "c5", "dense_hits", "binary"])
def test_compiled_tables_are_byte_identical_to_the_oracle(workloads, name):
    w = workloads[name]
    o = ob.Oracle(w.pattern_file)
    h = api.PFAC.createHostOnly()
    h.readPatternFromFile(w.pattern_file)
    info = h.info()
    assert (info.numOfPatterns, info.numOfStates, info.initialState, info.maxPatternLen, info.numOfLeaves) == (
        o.num_patterns, o.num_states, o.initial_state, o.max_pattern_len, o.num_leaves)
    assert info.numOfTableEntry == o.num_states * 256 and info.sizeOfTableEntry == 4
    assert np.array_equal(h.table(api.PFACX_TABLE_DENSE), o.dense_table())
    h.setPerfMode(api.PFAC_SPACE_DRIVEN)
    info = h.info()
    assert info.numOfTableEntry == o.hash_total and info.sizeOfTableEntry == 8
    assert np.array_equal(h.table(api.PFACX_TABLE_HASH_ROWPTR), o.hash_row())
    assert np.array_equal(h.table(api.PFACX_TABLE_HASH_VALPTR), o.hash_val())
    assert np.array_equal(h.table(api.PFACX_TABLE_INITIAL_ROW), o.initial_row())
    h.destroy()


def test_dump_transition_table_matches_user_guide_and_oracle(workloads, golden_dir, tmp_path):
    h = api.PFAC.createHostOnly()
    h.readPatternFromFile(os.path.join(golden_dir, "example_pattern"))
    out = tmp_path / "t.txt"
    h.dumpTransitionTable(str(out))
    assert out.read_bytes() == open(os.path.join(golden_dir, "userguide_r1.2_p21_table.txt"), "rb").read()
    for name in ("ex2", "binary", "c2"):
        h.readPatternFromFile(workloads[name].pattern_file)
        h.dumpTransitionTable(str(out))
        ref = tmp_path / "o.txt"
        ob.Oracle(workloads[name].pattern_file).dump_table(str(ref))
        assert out.read_bytes() == ref.read_bytes(), name
    h.destroy()


@pytest.mark.parametrize("platform", [api.PFAC_PLATFORM_CPU, api.PFAC_PLATFORM_CPU_OMP])
@pytest.mark.parametrize("perf", [api.PFAC_TIME_DRIVEN, api.PFAC_SPACE_DRIVEN])
def test_cpu_platforms_equal_oracle(workloads, oracle_results, platform, perf, monkeypatch):
    monkeypatch.setenv("OMP_NUM_THREADS", "4")          # CPU_OMP degrades to 1 thread without it (PFAC.cpp:904-908)
    for name in ("c1", "ex2", "c2", "c3", "dense_hits", "binary"):
        h = api.PFAC.createHostOnly()
        h.setPlatform(platform)
        h.setPerfMode(perf)
        h.readPatternFromFile(workloads[name].pattern_file)
        got = h.match_host_array(workloads[name].data)
        h.destroy()
        assert np.array_equal(got, oracle_results[name]), name


def test_host_reduce_on_cpu_platform(workloads, oracle_results):
    """ref the CPU branch of PFAC_matchFromHostReduce, PFAC.cpp:1036-1068."""
    w = workloads["c3"]
    h = api.PFAC.createHostOnly()
    h.readPatternFromFile(w.pattern_file)
    res = np.zeros(w.data.size, dtype=np.int32)
    pos = np.zeros(w.data.size, dtype=np.int32)
    st, count = h.matchFromHostReduce(w.data.ctypes.data, w.data.size, res.ctypes.data, pos.ctypes.data)
    h.destroy()
    ids, want_pos = ob.reduce(oracle_results["c3"])
    assert count == ids.size
    assert np.array_equal(res[:count], ids) and np.array_equal(pos[:count], want_pos)


def test_pattern_file_quirks(tmp_path):
    """last line without newline is dropped; a blank line before a pattern is rejected
    with a status (the reference asserts); duplicate lines load (test_duplicate_patterns...); CRLF keeps the CR."""
    h = api.PFAC.createHostOnly()

    def load(b):
        p = tmp_path / "q.pat"
        p.write_bytes(b)
        return h.readPatternFromFile(str(p), check=False)

    assert load(b"AB\nCD\nEF") == 0 and h.info().numOfPatterns == 2
    assert load(b"AB\n\nCD\n") == api.STATUS.INVALID_PARAMETER
    buf = np.zeros(4, dtype=np.int32)
    assert h.matchFromHost(buf.ctypes.data, 4, buf.ctypes.data, check=False) == api.STATUS.PATTERNS_NOT_READY, \
        "a failed load leaves no patterns behind (ref PFAC.cpp:678-681)"
    assert load(b"AB\nCD\nAB\n") == 0 and h.info().numOfPatterns == 3
    assert load(b"AB\r\nC\r\n") == 0 and h.info().maxPatternLen == 3
    assert load(b"") == 0 and h.info().numOfPatterns == 0
    got = h.match_host_array(np.frombuffer(b"anything", dtype=np.uint8))
    assert not got.any()
    h.destroy()


from tests.filter_model import prefilter_model, reduce_filter_model       # noqa: E402


def test_strict_and_crlf_readers(tmp_path):
    """PFACX_readPatternFromFileEx / ...MemoryEx: a last line without a newline is dropped like in the
    reference (PFAC_reorder_Table.cpp:181-195) but reported, or refused with PFACX_READ_STRICT; PFACX_READ_STRIP_CR takes
    "\\r\\n" line ends (the reference keeps the CR, user guide r1.2 p.15); flags 0 == the plain readers."""
    h = api.PFAC.createHostOnly()
    try:
        assert h.readPatternFromMemory(b"AB\nCD\nEF") == 0
        assert h.info().numOfPatterns == 2 and h.info().trailingBytesIgnored == 2
        assert h.readPatternFromMemoryEx(b"AB\nCD\nEF", api.PFACX_READ_STRICT, check=False) == api.STATUS.INVALID_PARAMETER
        buf = np.zeros(4, dtype=np.int32)
        assert h.matchFromHost(buf.ctypes.data, 4, buf.ctypes.data, check=False) == api.STATUS.PATTERNS_NOT_READY     # a refused load leaves nothing behind
        assert h.readPatternFromMemoryEx(b"AB\nCD\nEF\n", api.PFACX_READ_STRICT) == 0
        assert h.info().numOfPatterns == 3 and h.info().trailingBytesIgnored == 0
        # CRLF: without the flag the CR belongs to the pattern (and "EF\r" does not match "EF "), with it the set is {AB, C\rD, EF}
        crlf = b"AB\r\nC\rD\r\nEF\r\n"
        data = np.frombuffer(b"xxAByyC\rDzzEF EF\rq", dtype=np.uint8)
        assert h.readPatternFromMemory(crlf) == 0 and h.info().maxPatternLen == 4
        assert list(np.flatnonzero(h.match_host_array(data))) == [14]                  # only "EF\r"
        assert h.readPatternFromMemoryEx(crlf, api.PFACX_READ_STRIP_CR) == 0 and h.info().maxPatternLen == 3
        got = h.match_host_array(data)
        assert list(np.flatnonzero(got)) == [2, 6, 11, 14] and list(got[[2, 6, 11, 14]]) == [1, 2, 3, 3]
        assert h.readPatternFromMemoryEx(crlf[:-1], api.PFACX_READ_STRIP_CR | api.PFACX_READ_STRICT, check=False) == api.STATUS.INVALID_PARAMETER
        assert h.readPatternFromMemoryEx(b"AB\n", 4, check=False) == api.STATUS.INVALID_PARAMETER                # unknown flag
        # the file reader takes the same flags
        p = tmp_path / "crlf.pat"
        p.write_bytes(crlf + b"GH")
        assert h.readPatternFromFileEx(str(p), api.PFACX_READ_STRIP_CR) == 0
        assert h.info().numOfPatterns == 3 and h.info().trailingBytesIgnored == 2
        assert h.readPatternFromFileEx(str(p), api.PFACX_READ_STRICT, check=False) == api.STATUS.INVALID_PARAMETER
        assert h.readPatternFromFileEx(str(tmp_path / "missing"), 0, check=False) == api.STATUS.FILE_OPEN_ERROR
    finally:
        h.destroy()


@pytest.mark.parametrize("name", ["c1", "ex2", "c2", "c3", "c5", "dense_hits", "binary"])
def test_prefilter_has_no_false_negatives(workloads, oracle_results, name):
    """Every position with a non-zero result passes level 1 and the prefix ladder as the kernel evaluates them."""
    w = workloads[name]
    h = api.PFAC.createHostOnly()
    h.readPatternFromFile(w.pattern_file)
    info = h.info()
    level1, cand, walk = prefilter_model(h, w.data)
    h.destroy()
    hit = oracle_results[name] != 0
    assert np.all(level1[hit]) and np.all(cand[hit]) and np.all(walk[hit])
    assert info.filterHasShort == int(any(len(p) < 3 for p in open(w.pattern_file, "rb").read().split(b"\n") if p))
    assert info.filterLog2Bits <= 18
    lds = 32768 + ((1 << info.filterLog2BitsLadder) + (1 << info.filterLog2BitsFinal3)) // 8 + 8192 * info.filterHasShort    # level 1 has its 32 KiB whatever its size
    assert lds <= 97 * 1024, "the bitmaps share the LDS budget of the kernel (pfac_context.h: kFilterLdsBudget)"


@pytest.mark.parametrize("name", ["c1", "ex2", "c2", "c3", "c5", "dense_hits", "binary"])
def test_compacted_output_filter_has_no_false_negatives(workloads, oracle_results, name):
    """The compacted-output kernel has its own level 1 (gram1: one bit per 3-gram, 64 KiB) and depth-4 test (prefix4): every
    position with a non-zero result passes both as the kernel evaluates them, and on the Snort-style sample gram1 lets no
    more positions through than the two-bit gram3 of half its size (within a few per cent on this 1 MiB sample; 51.9 M against 53.2 M on the 1 GiB stream)."""
    w = workloads[name]
    h = api.PFAC.createHostOnly()
    h.readPatternFromFile(w.pattern_file)
    level1, walk = reduce_filter_model(h, w.data)
    full_level1 = prefilter_model(h, w.data)[0]
    h.destroy()
    hit = oracle_results[name] != 0
    assert np.all(level1[hit]) and np.all(walk[hit])
    if name == "c3":
        assert level1.sum() <= 1.05 * full_level1.sum()


def test_prefix_ladder_of_a_very_large_pattern_set(tmp_path):
    """120 000 patterns of 12..24 bytes over a 16-letter alphabet: more ladder nodes than the LDS bitmap may hold at a
    fifth full.  The compiler then gives up the extra level behind thin nodes and raises the thin threshold (fewer,
    shallower nodes); the filter must still pass every position the CPU platform (which never looks at the filter)
    reports."""
    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b"abcdefghijklmnop", dtype=np.uint8)
    pats = {alpha[rng.integers(0, 16, size=int(rng.integers(12, 25)))].tobytes() for _ in range(120000)}
    p = tmp_path / "big.pat"
    p.write_bytes(b"\n".join(sorted(pats)) + b"\n")
    h = api.PFAC.createHostOnly()
    h.setPerfMode(api.PFAC_SPACE_DRIVEN)                     # the dense table of this set would be > 1 GB
    h.readPatternFromFile(str(p))
    info = h.info()
    assert info.numOfPatterns == len(pats)
    assert info.ladderExtend == 0 and info.ladderThin >= 1 and 5 * (2 * info.ladderStops + info.ladderGoOns) <= (1 << info.filterLog2BitsLadder) * 1.0001 \
        or info.ladderThin >= (1 << 30)
    data = alpha[rng.integers(0, 16, size=400000)].copy()
    plist = sorted(pats)
    for k in range(2000):                                       # random text alone would hardly ever match a 12-byte pattern
        q = np.frombuffer(plist[int(rng.integers(0, len(plist)))], dtype=np.uint8)
        at = int(rng.integers(0, data.size - 32))
        data[at:at + q.size] = q
    want = h.match_host_array(data)
    level1, cand, walk = prefilter_model(h, data)
    h.destroy()
    hit = want != 0
    assert hit.sum() > 1000 and np.all(level1[hit]) and np.all(cand[hit]) and np.all(walk[hit])


def test_prefix_ladder_prunes_the_snort_style_stream(workloads, oracle_results):
    """The ladder is why the bench workload walks few positions: on the C3 sample fewer than half of the candidates
    (level-1 hits whose first four bytes are a pattern prefix) survive it, and every true match does."""
    w = workloads["c3"]
    h = api.PFAC.createHostOnly()
    h.readPatternFromFile(w.pattern_file)
    info = h.info()
    level1, cand, walk = prefilter_model(h, w.data)
    h.destroy()
    assert info.ladderStops > 0 and info.ladderGoOns > 0 and info.ladderThin == 1 and info.ladderExtend == 1
    assert walk.sum() < 0.5 * cand.sum() and np.all(walk[oracle_results["c3"] != 0])


def test_read_pattern_from_memory_equals_read_from_file(workloads):
    """PFACX_readPatternFromMemory compiles the same tables and gives the same
    results as PFAC_readPatternFromFile on the same bytes; error statuses match too."""
    for name in ("c1", "c3"):
        w = workloads[name]
        raw = open(w.pattern_file, "rb").read()
        a, b = api.PFAC.createHostOnly(), api.PFAC.createHostOnly()
        try:
            a.readPatternFromFile(w.pattern_file)
            b.readPatternFromMemory(raw)
            ia, ib = a.info(), b.info()
            assert (ia.numOfPatterns, ia.numOfStates, ia.maxPatternLen) == (ib.numOfPatterns, ib.numOfStates, ib.maxPatternLen)
            for which in (api.PFACX_TABLE_DENSE, api.PFACX_TABLE_INITIAL_ROW, api.PFACX_TABLE_FILTER_GRAM3):
                assert np.array_equal(a.table(which), b.table(which))
            sample = w.data[:20000]
            assert np.array_equal(a.match_host_array(sample), b.match_host_array(sample))
            b.readPatternFromMemory(b"second\nset\n")                     # replaces the first set
            assert b.info().numOfPatterns == 2
        finally:
            a.destroy()
            b.destroy()
    h = api.PFAC.createHostOnly()
    try:
        assert h.readPatternFromMemory(b"AB\n\nCD\n", check=False) == api.STATUS.INVALID_PARAMETER     # blank line
        assert h.readPatternFromMemory(b"AB\nAB\n", check=False) == api.STATUS.SUCCESS               # duplicate lines = one pattern
        assert h.readPatternFromMemory(b"AB\nCD", check=False) == api.STATUS.SUCCESS and h.info().numOfPatterns == 1   # bytes after the last newline are ignored
    finally:
        h.destroy()


@pytest.mark.parametrize("perf", [api.PFAC_TIME_DRIVEN, api.PFAC_SPACE_DRIVEN])
@pytest.mark.parametrize("name", ["c1", "ex2", "c3", "c5", "binary"])
def test_compiled_set_round_trip_on_the_host(workloads, oracle_results, tmp_path, name, perf):
    """PFACX_saveCompiled / PFACX_loadCompiled: the loaded handle has the same tables, the same
    facts and -- on the CPU platforms -- the same results; damaged or foreign files are refused with a status."""
    w = workloads[name]
    a = api.PFAC.createHostOnly()
    a.setPerfMode(perf)
    a.readPatternFromFile(w.pattern_file)
    path = str(tmp_path / "set.pfacx")
    a.saveCompiled(path)
    b = api.PFAC.createHostOnly()
    b.loadCompiled(path)
    ia, ib = a.info(), b.info()
    for field in ("numOfPatterns", "numOfStates", "initialState", "maxPatternLen", "numOfLeaves", "perfMode",
                  "numOfTableEntry", "sizeOfTableInBytes", "filterLog2Bits", "filterHasShort", "filterBitsSet", "filterLog2BitsLadder",
                  "filterBitsSetLadder", "ladderStops", "ladderGoOns", "ladderThin", "ladderExtend"):
        assert getattr(ia, field) == getattr(ib, field), field
    tables = [api.PFACX_TABLE_INITIAL_ROW, api.PFACX_TABLE_FILTER_GRAM3, api.PFACX_TABLE_FILTER_SHORT,
              api.PFACX_TABLE_FILTER_LADDER, api.PFACX_TABLE_FILTER_FINAL3]
    tables += [api.PFACX_TABLE_DENSE] if perf == api.PFAC_TIME_DRIVEN else [api.PFACX_TABLE_HASH_ROWPTR, api.PFACX_TABLE_HASH_VALPTR]
    for which in tables:
        assert np.array_equal(a.table(which), b.table(which)), which
    for platform in (api.PFAC_PLATFORM_CPU, api.PFAC_PLATFORM_CPU_OMP):
        b.setPlatform(platform)
        assert np.array_equal(b.match_host_array(w.data), oracle_results[name])
    da, db = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    a.dumpTransitionTable(da)
    b.dumpTransitionTable(db)
    assert open(da, "rb").read() == open(db, "rb").read()
    # a second load replaces the first; setPerfMode after a load rebuilds from the loaded trie
    b.loadCompiled(path)
    b.setPerfMode(api.PFAC_SPACE_DRIVEN if perf == api.PFAC_TIME_DRIVEN else api.PFAC_TIME_DRIVEN)
    assert np.array_equal(b.match_host_array(w.data), oracle_results[name])
    # refused: missing file, truncated file, flipped payload byte, wrong magic
    raw = open(path, "rb").read()
    assert b.loadCompiled(str(tmp_path / "nope.pfacx"), check=False) == api.STATUS.FILE_OPEN_ERROR
    for bad in (raw[: len(raw) // 2], raw[:100] + bytes([raw[100] ^ 0x40]) + raw[101:], b"NOTPFACX" + raw[8:], b""):
        open(path, "wb").write(bad)
        assert b.loadCompiled(path, check=False) == api.STATUS.INVALID_PARAMETER
        # a refused file leaves the handle as it was: same perf mode, same patterns, same results
        assert b.info().perfMode == (api.PFAC_SPACE_DRIVEN if perf == api.PFAC_TIME_DRIVEN else api.PFAC_TIME_DRIVEN)
        assert np.array_equal(b.match_host_array(w.data), oracle_results[name])
    assert a.saveCompiled(str(tmp_path / "no_such_dir" / "x"), check=False) == api.STATUS.FILE_OPEN_ERROR
    a.destroy()
    b.destroy()


def _fnv1a64(b):
    h = 0xcbf29ce484222325
    for x in b:
        h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def _sections(payload):
    """[(tag, offset of the data, bytes)] of a compiled set's payload (tag u32, size u64, data)"""
    out, at = [], 0
    while at + 12 <= len(payload):
        tag = int.from_bytes(payload[at:at + 4], "little")
        size = int.from_bytes(payload[at + 4:at + 12], "little")
        out.append((tag, at + 12, size))
        at += 12 + size
    return out


def test_crafted_compiled_sets_with_a_valid_checksum_are_refused(workloads, oracle_results, tmp_path):
    """The checksum of a compiled set is FNV-1a: anyone can recompute it.  So nothing a kernel indexes memory with is taken from
    the file -- every transition table is rebuilt from the trie at load -- and the trie itself is checked to BE one: a file
    whose edges form a cycle (a walk that never ends would run past the margin the kernels keep behind the input), point at
    the initial state, carry the same byte twice, or put a final state at the wrong depth is PFAC_STATUS_INVALID_PARAMETER,
    checksum recomputed or not.  Files of format 6 (tables inside) are refused by their version."""
    w = workloads["c3"]
    a = api.PFAC.createHostOnly()
    a.setPerfMode(api.PFAC_SPACE_DRIVEN)
    a.readPatternFromFile(w.pattern_file)
    path = str(tmp_path / "set.pfacx")
    a.saveCompiled(path)
    raw = bytearray(open(path, "rb").read())
    head, payload = raw[:40], raw[40:]                             # magic 8, version / fingerprint / perfMode / jumpLog2 4 x 4, payload bytes 8, FNV 8
    assert int.from_bytes(head[8:12], "little") == 7 and int.from_bytes(head[24:32], "little") == len(payload)
    secs = {tag: (off, size) for tag, off, size in _sections(payload)}
    assert 14 not in secs and 15 not in secs and 16 not in secs    # kSecHashRow, kSecHashVal, kSecChain: no transition table in the file
    off_next, size_next = secs[8]                                  # kSecEdgeNext
    off_ch, _ = secs[7]                                            # kSecEdgeCh
    off_begin, _ = secs[6]                                         # kSecEdgeBegin
    info = a.info()
    begin = np.frombuffer(bytes(payload[off_begin:off_begin + 4 * (info.numOfStates + 1)]), dtype=np.int32)
    nxt = np.frombuffer(bytes(payload[off_next:off_next + size_next]), dtype=np.int32)
    init = info.initialState
    e0 = int(begin[init])                                          # first edge of the initial state
    child = int(nxt[e0])
    grand_edge = int(begin[child])                                 # first edge of that child

    def crafted(mutate):
        p2 = bytearray(payload)
        mutate(p2)
        h2 = bytearray(head)
        h2[32:40] = _fnv1a64(bytes(p2)).to_bytes(8, "little")
        out = str(tmp_path / "crafted.pfacx")
        open(out, "wb").write(bytes(h2) + bytes(p2))
        return out

    def set_next(p2, edge, state):
        p2[off_next + 4 * edge:off_next + 4 * edge + 4] = int(state).to_bytes(4, "little", signed=True)

    b = api.PFAC.createHostOnly()
    b.readPatternFromFile(workloads["c1"].pattern_file)            # what must survive every refused load
    before = b.match_host_array(workloads["c1"].data)
    cases = {
        "a cycle: the child's first edge leads back to the child": lambda p2: set_next(p2, grand_edge, child),
        "an edge into the initial state": lambda p2: set_next(p2, grand_edge, init),
        "two edges into one state": lambda p2: set_next(p2, e0 + 1, child),
        "the same byte twice in one state": lambda p2: p2.__setitem__(off_ch + e0 + 1, p2[off_ch + e0]),
        "a state number beyond the trie": lambda p2: set_next(p2, grand_edge, info.numOfStates + 5),
    }
    for what, mutate in cases.items():
        assert b.loadCompiled(crafted(mutate), check=False) == api.STATUS.INVALID_PARAMETER, what
        assert np.array_equal(b.match_host_array(workloads["c1"].data), before), what
    # the untouched payload with its checksum recomputed loads, and matches like the set it was saved from
    assert b.loadCompiled(crafted(lambda p2: None), check=False) == api.STATUS.SUCCESS
    assert np.array_equal(b.match_host_array(w.data), oracle_results["c3"])
    assert b.info().trailingBytesIgnored == a.info().trailingBytesIgnored
    # an older format is refused by its version number
    old = bytearray(raw)
    old[8:12] = (6).to_bytes(4, "little")
    open(path, "wb").write(bytes(old))
    assert b.loadCompiled(path, check=False) == api.STATUS.INVALID_PARAMETER
    a.destroy()
    b.destroy()


def test_trailing_bytes_survive_a_compiled_set(tmp_path):
    """A pattern file whose last line has no newline: the bytes are ignored like in the reference, PFACX_getInfo says how many,
    and a set compiled from it says the same after PFACX_saveCompiled / PFACX_loadCompiled (format 7 stores the count)."""
    p = tmp_path / "trail.pat"
    p.write_bytes(b"alpha\nbeta\ngamm")
    a = api.PFAC.createHostOnly()
    a.readPatternFromFile(str(p))
    assert a.info().numOfPatterns == 2 and a.info().trailingBytesIgnored == 4
    f = str(tmp_path / "trail.pfacx")
    a.saveCompiled(f)
    b = api.PFAC.createHostOnly()
    b.loadCompiled(f)
    assert b.info().trailingBytesIgnored == 4 and b.info().numOfPatterns == 2
    a.destroy()
    b.destroy()


def test_compiled_set_with_cleared_filter_bitmaps_still_matches(workloads, oracle_results, tmp_path):
    """The prefilter bitmaps in a compiled set are not believed either (round 4 advice: a stale or crafted file with a valid
    checksum could clear bits -- no out-of-bounds read, addresses are masked, but matches silently dropped, and the
    full-result path, gram3 / ladder from the file, could disagree with the compacted-output path, gram1 / prefix4 always
    rebuilt).  They are rebuilt from the checked trie at load: a file whose 3-gram, ladder, length-3 and 2-byte bitmaps are
    all zero -- checksum recomputed -- loads, reports the bitmaps a fresh compile would, and matches like it."""
    w = workloads["c3"]
    a = api.PFAC.createHostOnly()
    a.setPerfMode(api.PFAC_SPACE_DRIVEN)
    a.readPatternFromFile(w.pattern_file)
    path = str(tmp_path / "set.pfacx")
    a.saveCompiled(path)
    raw = bytearray(open(path, "rb").read())
    head, payload = raw[:40], raw[40:]
    cleared = 0
    for tag, off, size in _sections(payload):
        if tag in (10, 11, 12, 13):                                 # kSecGram3, kSecLadder, kSecFinal3, kSecShort
            payload[off:off + size] = bytes(size)
            cleared += 1
    assert cleared == 4
    head[32:40] = _fnv1a64(bytes(payload)).to_bytes(8, "little")
    open(path, "wb").write(bytes(head) + bytes(payload))
    b = api.PFAC.createHostOnly()
    b.loadCompiled(path)
    for which in (api.PFACX_TABLE_FILTER_GRAM3, api.PFACX_TABLE_FILTER_LADDER, api.PFACX_TABLE_FILTER_FINAL3, api.PFACX_TABLE_FILTER_SHORT):
        assert np.array_equal(a.table(which), b.table(which)) and (which == api.PFACX_TABLE_FILTER_SHORT or a.table(which).any())
    assert a.info().filterBitsSet == b.info().filterBitsSet and a.info().filterBitsSetLadder == b.info().filterBitsSetLadder
    assert np.array_equal(b.match_host_array(w.data), oracle_results["c3"])
    a.destroy()
    b.destroy()


def test_info_struct_is_versioned_by_its_size():
    """PFACX_info_t / PFACX_scan_stats_t carry their size: a caller built against an older, shorter header is never written past,
    a caller that forgot to set the size is refused."""
    import ctypes as C
    h = api.PFAC.createHostOnly()
    lib = api.load_library()
    info = api.PFACX_info()
    assert lib.PFACX_getInfo(h._h, C.byref(info)) == api.STATUS.INVALID_PARAMETER          # structSize == 0
    buf = (C.c_ubyte * C.sizeof(api.PFACX_info))(*([0xAB] * C.sizeof(api.PFACX_info)))
    short = C.cast(buf, C.POINTER(api.PFACX_info))
    short.contents.structSize = 48                                  # "an older header": the first 48 bytes only
    assert lib.PFACX_getInfo(h._h, short) == api.STATUS.SUCCESS
    assert short.contents.structSize == 48 and all(x == 0xAB for x in bytes(buf)[48:])
    full = h.info()
    assert full.structSize == C.sizeof(api.PFACX_info) and full.deviceTableBytes == 0 and full.hasDevice == 0
    h.destroy()


def test_sparse_fnv_equals_the_full_vector_fnv():
    """workloads.fnv1a_sparse_i32 (used to pin 4 GiB result vectors from their sparse form) == FNV-1a-64 of the vector."""
    from oracle import binding as ob
    from pfac_amd import workloads as wl
    rng = np.random.default_rng(3)
    for n, k in ((1, 0), (1, 1), (1000, 7), (300001, 5000)):
        v = np.zeros(n, dtype=np.int32)
        pos = np.sort(rng.choice(n, k, replace=False))
        v[pos] = rng.integers(1, 1 << 20, k)
        assert wl.fnv1a_sparse_i32(pos, v[pos], n) == ob.digest(v)[0] == wl.fnv1a(v)


@pytest.mark.parametrize("perf", [api.PFAC_TIME_DRIVEN, api.PFAC_SPACE_DRIVEN])
def test_duplicate_patterns_are_one_pattern_with_the_highest_id(tmp_path, perf):
    """Rule sets repeat lines.  The reference pushes a second edge for the same byte (dense table: the ID its
    unstable sort placed last, longer patterns below the first copy are lost, PFAC.cpp:376-381; hashed build:
    fails, :506-551).  Here the copies are one pattern reported under the highest of their IDs: the result equals
    the oracle's on the same file with the earlier copies replaced by patterns that cannot occur."""
    from oracle import binding as ob
    pats = [b"AB", b"CD", b"AB", b"ABX", b"CD", b"Q", b"CDE", b"Q", b"AB"]
    unique = [b"\x01\x02", b"\x01\x03", b"\x01\x04", b"ABX", b"CD", b"\x01\x05", b"CDE", b"Q", b"AB"]   # same IDs for the last copies
    data = np.frombuffer(b"xxABXyCDEzQABABXCDCDQ" * 50 + b"AB", dtype=np.uint8)
    fa, fb = tmp_path / "dup.pat", tmp_path / "uniq.pat"
    fa.write_bytes(b"".join(p + b"\n" for p in pats))
    fb.write_bytes(b"".join(p + b"\n" for p in unique))
    want = ob.Oracle(str(fb)).match(data)
    assert set(np.unique(want)) == {0, 4, 5, 7, 8, 9}
    h = api.PFAC.createHostOnly()
    h.setPerfMode(perf)
    h.readPatternFromFile(str(fa))
    assert h.info().numOfPatterns == len(pats)
    for platform in (api.PFAC_PLATFORM_CPU, api.PFAC_PLATFORM_CPU_OMP):
        h.setPlatform(platform)
        assert np.array_equal(h.match_host_array(data), want)
    h.destroy()


@pytest.mark.parametrize("name,perf", [("c1", "dense"), ("ex2", "hash"), ("c2", "hash"), ("c3", "hash"), ("c3", "dense"), ("c5", "hash"),
                                       ("dense_hits", "hash"), ("binary", "hash")])
def test_chained_table_walk_equals_oracle(workloads, oracle_results, name, perf):
    """The device-only chained table (PFACX_TABLE_CHAIN), walked the way the filter kernel's walkers do -- first
    slot from the jump table at hash(first four bytes), restart in the initial state's bucket if that slot is somebody
    else's, then one slot per transition with its chain -- gives the reference's result at every position.  No
    prefilter here: the tables alone must be exact."""
    w = workloads[name]
    h = api.PFAC.createHostOnly()
    h.setPerfMode(api.PFAC_TIME_DRIVEN if perf == "dense" else api.PFAC_SPACE_DRIVEN)
    h.readPatternFromFile(w.pattern_file)
    slots = h.table(api.PFACX_TABLE_CHAIN).reshape(-1, 4)
    info = h.info()
    h.destroy()
    J = info.chainJumpLog2
    assert 10 <= J <= 20 and info.chainSlots == len(slots) and len(slots) % 2 == 0
    ext_delta = len(slots) // 2                    # N slot headers, then N extension units: the unit of slot i at N + i
    jump_base = ext_delta - (2 << J)               # the jump table; behind it the LONG jump table (same hash, chains of up to 23 bytes)
    root_row = jump_base - 256
    EMPTY, FINAL, WIDE = 1 << 14, 1 << 13, 1 << 15         # pfac_context.h: kSlotEmpty, kSlotFinal, kSlotWide; a leaf has k == 0
    long_steps = 0

    def walk_all(stream, expect, long_jump=False):
        nonlocal data
        n = len(stream)
        data = bytes(stream) + bytes(80)
        limit = max(n - info.maxPatternLen, 0)    # beyond it a walk would read the padding
        used_jump = fell_back = 0
        for i in range(limit):
            x = int.from_bytes(data[i:i + 4], "little")
            match = 0
            ok, leaf, ident, row, ks, used = step(jump_base + (long_jump << J) + (((x * 0x9E3779B1) & 0xFFFFFFFF) >> (32 - J)), long_jump, data[i], i + 1)
            if ok:
                used_jump += 1
            else:                                  # restart in the initial state's bucket (k = 128, S = 256: the byte itself)
                fell_back += 1
                ok, leaf, ident, row, ks, used = step(root_row + data[i], False, data[i], i + 1)
            depth = 0
            while ok:
                if ident:
                    match = ident
                depth += used
                if leaf:
                    break
                b0 = data[i + depth]
                r = ((((ks >> 16) & 0xFF) * b0) >> 7) & (ks >> 24)        # pfac_context.h: chainSlotOf
                ok, leaf, ident, row, ks, used = step(row + r, bool(ks & WIDE), b0, i + depth + 1)   # only the slots of WIDE buckets may be long
            assert match == int(expect[i]), (name, perf, i, match, int(expect[i]))
        return used_jump, fell_back

    data = b""

    def step(at, wide, b0, p):
        """transition through the slot at index `at` (wide: the slot that led here says its bucket may hold long slots) on edge
        byte b0 with the input behind it at p: (ok, leaf, match id or 0, end row, ks, bytes consumed)"""
        nonlocal long_steps
        slot = slots[at]
        meta = int(slot[0])
        ln = (meta >> 8) & 0x1F
        chain = int(slot[2]).to_bytes(4, "little") + int(slot[3]).to_bytes(4, "little")
        if ln > 7:                                 # only slots of wide buckets fold more than 7 bytes: 8 in the header, the rest in the unit
            assert wide and ln <= 23
            chain += b"".join(int(v).to_bytes(4, "little") for v in slots[at + ext_delta])
        else:
            assert not slots[at + ext_delta].any()     # the unit of a short slot is never written
            long_steps += 1
        ok = (meta & (EMPTY | 0xFF)) == b0 and chain[:ln] == data[p:p + ln]
        leaf = (meta >> 16) & 0xFF == 0
        ident = 0
        if ok and meta & FINAL:
            ident = int(slot[1]) if leaf else int(slot[3])
        return ok, leaf, ident, int(slot[1]), meta, 1 + ln

    used_jump, fell_back = walk_all(w.data[:30000], oracle_results[name])
    used_long, _ = walk_all(w.data[:30000], oracle_results[name], long_jump=True)     # the long jump table: same results (a slot that
    assert used_long <= used_jump                                                    # compares more bytes sends more walks back to the root)
    if name in ("c2", "c3"):
        assert used_jump > 0                       # the stream does contain 4-byte pattern prefixes
    if name == "c5":
        assert long_steps > 0                      # the near-miss stream walks the long single-successor runs: wide buckets
    if name == "c2":
        # 1000 random prefixes in 8192 slots: some collide.  A pattern whose prefix lost its slot must be found
        # through the restart: a stream of exactly those patterns, against the oracle
        from oracle import binding as ob
        pats = [p for p in open(w.pattern_file, "rb").read().split(b"\n") if p]
        lost = [p for p in pats if len(p) >= 4 and
                (int(slots[jump_base + (((int.from_bytes(p[:4], "little") * 0x9E3779B1) & 0xFFFFFFFF) >> (32 - J))][2]) & 0xFFFFFF)
                != int.from_bytes(p[1:4], "little")]
        assert lost, "every prefix has its own jump slot: the test lost its point"
        stream = np.frombuffer(b"\x00".join(lost) + bytes(64), dtype=np.uint8)
        o = ob.Oracle(w.pattern_file, hashed=False)
        want = o.match(stream)
        o.close()
        assert np.count_nonzero(want) >= len(lost)
        _, fell_back2 = walk_all(stream, want)
        assert fell_back2 >= len(lost)


@pytest.mark.parametrize("workers", [1, 3, 8])
def test_multi_worker_calls_on_a_cpu_platform_handle(workloads, oracle_results, workers):
    """PFACX_matchFromHostMultiGPU / PFACX_matchFromHostReduceMultiGPU on a host-only handle: the `numDevices` workers run as host threads over
    the CPU matchers -- the slices (boundaries on 1 KiB tiles), the maxPatternLen read-ahead behind each, the rebasing of every worker's
    positions and the moving-together of the lists are the GPU workers' (multi_gpu.cpp), so eight workers can be exercised without a GPU
    (omp_PFAC.cpp:257-394 is the reference's user-side version).  Matches planted across every slice boundary; full vector and pairs == oracle."""
    from oracle import binding as ob
    w = workloads["c2"]
    data = w.data.copy()
    n = int(data.size)
    pats = [p for p in open(w.pattern_file, "rb").read().split(b"\n") if len(p) >= 12]
    for i in range(1, workers):
        b = (n * i // workers) // 1024 * 1024
        p = np.frombuffer(pats[i % len(pats)], dtype=np.uint8)
        data[b - p.size // 2:b - p.size // 2 + p.size] = p                  # begins in worker i - 1's slice, ends in worker i's
    o = ob.Oracle(w.pattern_file)
    want = o.match(data)
    o.close()
    for i in range(1, workers):
        b = (n * i // workers) // 1024 * 1024
        p = pats[i % len(pats)]
        assert want[b - len(p) // 2] != 0
    for perf in (api.PFAC_TIME_DRIVEN, api.PFAC_SPACE_DRIVEN):
        h = api.PFAC.createHostOnly()
        try:
            h.setPerfMode(perf)
            h.readPatternFromFile(w.pattern_file)
            got = np.full(n, -9, dtype=np.int32)
            h.matchFromHostMultiGPU(data.ctypes.data, n, got.ctypes.data, devices=list(range(workers)))
            assert np.array_equal(got, want), (perf, workers, int(np.flatnonzero(got != want)[0]))
            ids, pos = np.full(n, -9, dtype=np.int32), np.full(n, -9, dtype=np.int32)
            _, count = h.matchFromHostReduceMultiGPU(data.ctypes.data, n, ids.ctypes.data, pos.ctypes.data, devices=list(range(workers)))
            nz = np.flatnonzero(want)
            assert count == nz.size and np.array_equal(pos[:count], nz) and np.array_equal(ids[:count], want[nz]), (perf, workers, count, nz.size)
        finally:
            h.destroy()

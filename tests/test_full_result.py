"""PFAC_matchFromDevice on the GPU == the oracle, bit for bit, through the C ABI: every BASELINE workload at oracle size in every table
mode and kernel variant, ragged sizes, tile / chunk boundaries, fuzzed pattern sets, misaligned pointers, the API's life cycle.

Modelled on the reference's example programs: PFAC/test/simple_example.cpp (matchFromHost), README.md example 2 (matchFromDevice)
and the only self-checking reference test, PFAC/test/omp_PFAC.cpp:396-439 (sliced run == single run); PFAC/src/PFAC_kernel.cu:
102-108, 301-345 (patterns longer than 511 bytes).  Device memory comes from torch; the match itself never does."""

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


@pytest.mark.parametrize("perf,tex,mode_name", MODES)
@pytest.mark.parametrize("variant,variant_name", VARIANTS)
@pytest.mark.parametrize("name", ["c1", "ex2", "c2", "c3", "c5", "dense_hits", "binary"])
def test_match_from_device_equals_oracle(workloads, oracle_results, name, perf, tex, mode_name, variant, variant_name):
    w = workloads[name]
    h = make_handle(w.pattern_file, perf, tex, variant)
    try:
        got = device_match(h, w.data)
    finally:
        h.destroy()
    assert_same(got, oracle_results[name], f"{name}/{mode_name}/{variant_name}")


def test_readme_example_known_answer(golden_dir):
    """README.md:113-120 through matchFromHost on the GPU platform (simple_example.cpp)."""
    import json, os
    ka = json.load(open(os.path.join(golden_dir, "known_answers.json")))["example1"]
    data = np.fromfile(os.path.join(golden_dir, ka["input_file"]), dtype=np.uint8)
    for perf in (api.PFAC_TIME_DRIVEN, api.PFAC_SPACE_DRIVEN):
        h = api.PFAC.create()
        h.setPerfMode(perf)
        h.readPatternFromFile(os.path.join(golden_dir, ka["pattern_file"]))
        got = h.match_host_array(data)
        h.destroy()
        assert got.tolist() == ka["result_full"]
        pos = np.nonzero(got)[0]
        assert pos.tolist() == ka["reduce_pos"] and got[pos].tolist() == ka["reduce_id"]


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 63, 64, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 16385, 65537])
def test_ragged_sizes(workloads, n):
    """Tile / dword / tail boundaries of the vector path (H6 in SURVEY.md): matches that end
    exactly at n, start in the last bytes, or would need bytes beyond n."""
    from oracle import binding as ob
    w = workloads["dense_hits"]
    reps = (n // w.data.size) + 2
    data = np.tile(w.data, reps)[17:17 + n].copy()
    o = ob.Oracle(w.pattern_file, hashed=False)
    want = o.match(data)
    o.close()
    for perf, tex, mode_name in MODES[::2]:
        h = make_handle(w.pattern_file, perf, tex)
        try:
            assert_same(device_match(h, data), want, f"n={n}/{mode_name}")
        finally:
            h.destroy()


@pytest.mark.parametrize("perf,tex,mode_name", MODES)
def test_long_walks_across_every_tile_and_chunk_boundary(workdir, perf, tex, mode_name):
    """Full matches, prefixes-that-are-patterns and near misses of 60-byte patterns planted so that
    they straddle a 2 KiB chunk (and 1 KiB tile, 16-byte lane) boundary at every offset 0..71:
    exercises the walk queue's entries (36-byte windows of the register-window walker, {buffer, offset} codes of the stage
    walker) across lanes / tiles / chunks, chains longer than a slot header (long slots and their units), window re-fetches,
    walks that run off their staged chunk, and the bounded walks over the ends of the input."""
    import os
    from oracle import binding as ob
    from pfac_amd import workloads as wl
    rng = np.random.Generator(np.random.PCG64(2024))
    long_a = bytes(rng.integers(97, 123, 60, dtype=np.uint8))          # one long single-successor chain
    long_b = long_a[:31] + bytes(rng.integers(65, 91, 29, dtype=np.uint8))   # shares 31 bytes, then diverges
    pats = [long_a, long_b, long_a[:9], long_a[:17], long_b[:40],      # patterns that are prefixes of patterns
            long_a[5:25], b"zq", b"zqx" * 6]
    pf = wl.write_pattern_file(os.path.join(workdir, "longwalk.pat"), pats)
    n = 2048 * 80 + 777
    data = rng.integers(0, 4, n, dtype=np.uint8) + 48                   # filler that matches nothing
    plant = [long_a, long_b, long_a[:59], long_a[:30] + b"#", long_b[:45], long_a[:16], long_a[5:24]]
    for j in range(72):
        at = 2048 * (3 + j) - j
        p = plant[j % len(plant)]
        data[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
    tail = np.frombuffer(long_a, dtype=np.uint8)
    data[n - 60:] = tail                                                # a match that ends exactly at n
    data[n - 200:n - 141] = tail[:59]                                   # a near miss inside the tail range
    o = ob.Oracle(pf, hashed=False)
    want = o.match(data)
    o.close()
    assert np.count_nonzero(want) > 100
    h = make_handle(pf, perf, tex)
    try:
        assert_same(device_match(h, data), want, f"long walks/{mode_name}")
    finally:
        h.destroy()


@pytest.mark.parametrize("seed", range(16))
def test_fuzzed_pattern_sets_over_tiny_alphabets(workdir, seed):
    """Random pattern sets over 2-4 symbol alphabets: patterns that are prefixes of patterns at every
    depth (final states with successors, the pattern-ID-in-chain encoding and its chain cut), long
    single-successor chains, 1- and 2-byte patterns (exact short bitmap), bytes 0x00 / 0xFF, and an
    input in which almost every position walks.  All four table modes against the oracle."""
    import os
    from oracle import binding as ob
    from pfac_amd import workloads as wl
    rng = np.random.Generator(np.random.PCG64(900 + seed))
    alphabet = [bytes([b]) for b in rng.choice([0x00, 0xFF, 0x41, 0x42, 0x7A, 0x20, 0x0D], size=int(rng.integers(2, 5)), replace=False)]
    pats = set()
    base = b"".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), 48))
    for cut in rng.integers(1, 48, int(rng.integers(3, 14))):           # prefixes of one long string
        pats.add(base[:int(cut)])
    while len(pats) < int(rng.integers(8, 70)):
        ln = int(rng.integers(1 if seed % 2 else 3, 41))
        pats.add(b"".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), ln)))
    pats = sorted(pats, key=lambda p: (rng.random(), p))               # file order = pattern IDs: shuffled
    pf = wl.write_pattern_file(os.path.join(workdir, f"fuzz{seed}.pat"), pats)
    n = int(rng.integers(40_000, 200_000))
    idx = rng.integers(0, len(alphabet), n)
    data = np.frombuffer(b"".join(alphabet), dtype=np.uint8)[idx].copy()
    at = int(rng.integers(0, n - 100))
    data[at:at + len(base)] = np.frombuffer(base, dtype=np.uint8)
    o = ob.Oracle(pf, hashed=False)
    want = o.match(data)
    o.close()
    for perf, tex, mode_name in MODES:
        h = make_handle(pf, perf, tex)
        try:
            assert_same(device_match(h, data), want, f"fuzz seed {seed}/{mode_name}")
            d_in = torch.from_numpy(data).to("cuda:0")
            d_res = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
            d_pos = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
            st, count = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_res.data_ptr(), d_pos.data_ptr())
            nz = np.nonzero(want)[0]
            assert count == nz.size and np.array_equal(d_pos[:count].cpu().numpy(), nz) and \
                np.array_equal(d_res[:count].cpu().numpy(), want[nz]), f"fuzz seed {seed}/{mode_name} reduce"
        finally:
            h.destroy()


@pytest.mark.parametrize("in_off,out_off", [(1, 0), (2, 0), (3, 0), (0, 1), (0, 2), (0, 3), (1, 1), (4, 4), (8, 0)])
def test_misaligned_pointers(workloads, oracle_results, in_off, out_off):
    """The reference casts the input to int* (PFAC_kernel.cu:203); this build accepts any alignment."""
    w = workloads["c3"]
    data = w.data[: 200001]
    from oracle import binding as ob
    o = ob.Oracle(w.pattern_file, hashed=False)
    want = o.match(data)
    o.close()
    h = make_handle(w.pattern_file, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_OFF)
    try:
        assert_same(device_match(h, data, in_off, out_off), want, f"offsets {in_off},{out_off}")
    finally:
        h.destroy()


def test_misaligned_pointers_stay_on_the_vector_kernel(workdir, capsys):
    """The reference casts the input to int* (PFAC_kernel.cu:203) and asks for a padded buffer (PFAC.cpp:838-842); this
    library takes any pointer.  Round 2 sent a call whose pointers were not 16-byte aligned to the simple kernel as a
    whole (18 times slower); now only the <= 15 positions in front of the first aligned input byte go there.  64 MiB of
    the Snort-style stream at several input / result offsets: results equal the aligned call's and the rate stays within a
    fifth of it.
    ANCHOR: this is a HIP-vs-HIP comparison at 64 MiB (the aligned call of the same handle is the expected value); what ties
    it to the reference is that aligned call's first MiB against the oracle, plus test_full_size_result_equals_reference_digest
    (the same stream, aligned, whole 1 GiB vector == SHA-256 of the reference's output)."""
    from oracle import binding as ob
    from pfac_amd import hiprt
    cfg = wl.make_config("c3")
    pf = wl.write_pattern_file(os.path.join(workdir, "misaligned.pat"), cfg.patterns)
    n = 64 << 20
    host = cfg.input_slice(n, 0)
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_AUTOMATIC, api.PFACX_KERNEL_AUTO)
    rates, ref = {}, None
    try:
        d_in = torch.zeros(n + 64, dtype=torch.uint8, device="cuda:0")
        d_out = torch.zeros(n + 64, dtype=torch.int32, device="cuda:0")
        for in_off, out_off in ((0, 0), (1, 0), (3, 1), (8, 2), (13, 3), (0, 1)):
            d_in[in_off:in_off + n].copy_(torch.from_numpy(host))
            d_out.fill_(-9)
            pi, po = d_in.data_ptr() + in_off, d_out.data_ptr() + 4 * out_off
            h.matchFromDevice(pi, n, po)
            torch.cuda.synchronize()
            a, b = hiprt.Event(), hiprt.Event()
            a.record(0)
            for _ in range(5):
                h.matchFromDevice(pi, n, po)
            b.record(0)
            torch.cuda.synchronize()
            rates[(in_off, out_off)] = round(n / (a.elapsed_ms(b) / 5 / 1e3) / 1e9, 1)
            got = d_out[out_off:out_off + n].cpu().numpy()
            assert int(d_out[out_off + n]) == -9 and (out_off == 0 or int(d_out[out_off - 1]) == -9), "wrote outside the result vector"
            if ref is None:
                o = ob.Oracle(pf, dense=False, hashed=True)
                want = o.match(host[: (1 << 20) + 256], hashed=True, omp=True)[: 1 << 20]
                o.close()
                assert np.array_equal(got[: 1 << 20], want)
                ref = got
            else:
                assert np.array_equal(got, ref), (in_off, out_off)
            # the compacted-output call on the same (misaligned) input: the positions in front of the first aligned byte
            # and the end of the input are walked inside the launch and join the list of pairs
            d_res = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
            d_pos = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
            _, count = h.matchFromDeviceReduce(pi, n, d_res.data_ptr(), d_pos.data_ptr())
            nz = np.nonzero(ref)[0]
            assert count == nz.size and np.array_equal(d_pos[:count].cpu().numpy(), nz) and np.array_equal(d_res[:count].cpu().numpy(), ref[nz]), (in_off, "reduce")
            del d_res, d_pos
    finally:
        h.destroy()
    with capsys.disabled():
        print("\n[64 MiB Snort-style, (input byte offset, result int offset) -> input GB/s]", rates)
    if perf_asserts():                                     # a comparison of event times: not part of a correctness run (gpu_helpers.perf_asserts)
        assert min(rates.values()) >= 0.8 * rates[(0, 0)], rates


def test_size_zero_and_argument_checks(workloads):
    """Status codes and check order of PFAC_matchFromDevice (ref PFAC.cpp:846-861)."""
    w = workloads["c1"]
    h = api.PFAC.create()
    buf = torch.zeros(64, dtype=torch.int32, device="cuda:0")
    assert h.matchFromDevice(buf.data_ptr(), 16, buf.data_ptr(), check=False) == api.STATUS.PATTERNS_NOT_READY
    h.readPatternFromFile(w.pattern_file)
    assert h.matchFromDevice(0, 16, buf.data_ptr(), check=False) == api.STATUS.INVALID_PARAMETER
    assert h.matchFromDevice(buf.data_ptr(), 16, 0, check=False) == api.STATUS.INVALID_PARAMETER
    buf.fill_(-3)
    assert h.matchFromDevice(buf.data_ptr(), 0, buf.data_ptr(), check=False) == api.STATUS.SUCCESS
    torch.cuda.synchronize()
    assert int((buf == -3).sum()) == 64, "size 0 must not write"
    assert h.info().hasDevice == 1
    h.destroy()


def test_set_perf_mode_after_load_rebuilds_tables(workloads, oracle_results):
    """ref PFAC_setPerfMode, PFAC.cpp:794-814."""
    w = workloads["c2"]
    h = make_handle(w.pattern_file, api.PFAC_TIME_DRIVEN, api.PFAC_AUTOMATIC)
    try:
        assert_same(device_match(h, w.data), oracle_results["c2"], "dense")
        h.setPerfMode(api.PFAC_SPACE_DRIVEN)
        assert h.info().sizeOfTableEntry == 8
        assert_same(device_match(h, w.data), oracle_results["c2"], "hash after switch")
        h.setPerfMode(api.PFAC_TIME_DRIVEN)
        assert_same(device_match(h, w.data), oracle_results["c2"], "dense after switch back")
        assert h.info().textureMode == api.PFAC_TEXTURE_ON, "AUTOMATIC resolves to ON below 2^27 entries (ref PFAC.cpp:819-833)"
    finally:
        h.destroy()


def test_reload_patterns_replaces_previous_set(workloads, oracle_results):
    """ref PFAC_readPatternFromFile, PFAC.cpp:663-666."""
    h = make_handle(workloads["c2"].pattern_file, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_OFF)
    try:
        h.readPatternFromFile(workloads["ex2"].pattern_file)
        assert h.info().numOfPatterns == 10
        assert_same(device_match(h, workloads["ex2"].data), oracle_results["ex2"], "after reload")
    finally:
        h.destroy()


def test_cpu_platforms_agree_with_gpu(workloads, oracle_results):
    """PFAC_setPlatform: same handle, CPU / CPU_OMP / GPU all give the oracle's answer."""
    w = workloads["c3"]
    h = make_handle(w.pattern_file, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_OFF)
    try:
        for platform in (api.PFAC_PLATFORM_GPU, api.PFAC_PLATFORM_CPU, api.PFAC_PLATFORM_CPU_OMP):
            h.setPlatform(platform)
            assert_same(h.match_host_array(w.data), oracle_results["c3"], f"platform {platform}")
    finally:
        h.destroy()


def test_slices_with_overlap_equal_single_call(workloads, oracle_results):
    """The reference's own self-check (omp_PFAC.cpp:319-439): chunks with a max_patternLen+1 tail,
    only [start,end) kept, must reproduce the single-call result."""
    from pfac_amd import sharding
    w = workloads["c3"]
    h = make_handle(w.pattern_file, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_ON)
    try:
        overlap = sharding.overlap_bytes(h.info().maxPatternLen)
        n = w.data.size
        got = np.empty(n, dtype=np.int32)
        for s in sharding.plan_slices(n, 7, overlap):
            part = device_match(h, w.data[s.start:s.read_end])
            got[s.start:s.end] = part[: s.end - s.start]
        assert_same(got, oracle_results["c3"], "7 slices")
    finally:
        h.destroy()


def test_two_handles_interleaved(workloads, oracle_results):
    """Two handles with different pattern sets used alternately (SimpleMultiGPU_pthread.cpp idea)."""
    a = make_handle(workloads["c2"].pattern_file, api.PFAC_TIME_DRIVEN, api.PFAC_TEXTURE_ON)
    b = make_handle(workloads["c5"].pattern_file, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_OFF)
    try:
        for _ in range(2):
            assert_same(device_match(a, workloads["c2"].data), oracle_results["c2"], "handle a")
            assert_same(device_match(b, workloads["c5"].data), oracle_results["c5"], "handle b")
    finally:
        a.destroy()
        b.destroy()


def test_golden_vectors_of_the_reference_build(golden_dir, workdir):
    """tests/golden/ref_vectors.json (produced by the reference's own CPU code) through matchFromDevice."""
    import json, os
    from pfac_amd import workloads as wl
    from tests.test_oracle_golden import _golden_input
    vec = json.load(open(os.path.join(golden_dir, "ref_vectors.json")))
    for case in vec["cases"]:
        pats = getattr(wl, case["patterns"]["fn"])(*case["patterns"]["args"])
        pf = wl.write_pattern_file(os.path.join(workdir, "gpu_golden_" + case["name"] + ".pat"), pats)
        data = _golden_input(wl, case, pats)
        for perf, tex, mode_name in MODES:
            h = make_handle(pf, perf, tex)
            try:
                got = device_match(h, data)
            finally:
                h.destroy()
            pos = np.nonzero(got)[0]
            assert pos.tolist() == case["positions"], f"{case['name']}/{mode_name}"
            assert got[pos].tolist() == case["ids"], f"{case['name']}/{mode_name}"


def test_cpp_example_program_prints_the_readme_answer(tmp_path):
    """examples/simple_example.cpp (re-authored PFAC/test/simple_example.cpp) linked against the
    drop-in library prints README.md:113-120."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "-B"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(root, "examples", "simple_example")], cwd=root, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("At position")]
    assert lines == ["At position    0, match pattern 1", "At position    1, match pattern 3",
                     "At position    2, match pattern 4", "At position    4, match pattern 4",
                     "At position    6, match pattern 2"]


@pytest.mark.parametrize("perf,tex,mode_name", MODES)
def test_patterns_longer_than_511_bytes(workdir, perf, tex, mode_name):
    """The reference has a separate code path for maxPatternLen > 511 (PFAC_kernel.cu:102-108, 301-345).  One
    600-byte and one 2 000-byte pattern (plus short ones), planted across chunk (2 KiB), span (8 KiB) and
    filter-kernel / tail-kernel boundaries, complete and with a wrong last byte."""
    from oracle import binding as ob
    rng = np.random.Generator(np.random.PCG64(600))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    p600 = alpha[rng.integers(0, 4, 600)].tobytes()
    p2000 = alpha[rng.integers(0, 4, 2000)].tobytes()
    pats = [p600, p2000, p600[:40] + b"N", b"ACGTNN", p2000[100:130]]
    pf = wl.write_pattern_file(os.path.join(workdir, "long.pat"), pats)
    n = (3 << 20) + 777
    data = np.frombuffer(b"N", dtype=np.uint8).repeat(n).copy()
    spots = [5, 2048 - 300, 8192 - 1000, (1 << 20) - 1999, (2 << 20) - 17, n - 2000, n - 2600, n - 4096 - 600, n - 2032 - 600 + 3]
    for k, at in enumerate(spots):
        p = np.frombuffer(p2000 if k % 2 else p600, dtype=np.uint8)
        at = min(at, n - p.size)
        data[at:at + p.size] = p
        if k % 3 == 2:
            data[at + p.size - 1] = ord("N")                     # near miss: walks the whole pattern, reports a shorter one or nothing
    o = ob.Oracle(pf)
    want = o.match(data, omp=True)
    o.close()
    assert set(np.unique(want)) >= {0, 1, 2}
    for variant in (api.PFACX_KERNEL_FILTER, api.PFACX_KERNEL_AUTO):
        h = make_handle(pf, perf, tex, variant)
        try:
            assert h.info().maxPatternLen == 2000
            assert_same(device_match(h, data), want, f"long patterns / {mode_name} / variant {variant}")
            got = np.full(n, -3, dtype=np.int32)
            h.matchFromHost(data.ctypes.data, n, got.ctypes.data)
            assert_same(got, want, f"long patterns / matchFromHost / {mode_name}")
        finally:
            h.destroy()


def test_patterns_from_memory_and_compiled_files_on_the_gpu(workloads, oracle_results, tmp_path):
    """PFACX_readPatternFromMemory and PFACX_saveCompiled / PFACX_loadCompiled feed the same
    kernels: results equal the oracle in both perf modes; a set saved by a host-only handle loads on the GPU."""
    for name in ("c3", "c5", "dense_hits"):
        w = workloads[name]
        raw = open(w.pattern_file, "rb").read()
        for perf, tex, mode_name in (MODES[1], MODES[2]):
            h = api.PFAC.create()
            h.setPerfMode(perf)
            h.setTextureMode(tex)
            h.setKernelVariant(api.PFACX_KERNEL_FILTER)
            h.readPatternFromMemory(raw)
            assert_same(device_match(h, w.data), oracle_results[name], f"{name}/{mode_name} patterns from memory")
            f1 = str(tmp_path / f"{name}_{mode_name}.pfacx")
            h.saveCompiled(f1)
            h.destroy()
            h2 = api.PFAC.create()
            h2.setKernelVariant(api.PFACX_KERNEL_FILTER)
            h2.setTextureMode(tex)
            h2.loadCompiled(f1)
            assert h2.info().perfMode == perf
            assert_same(device_match(h2, w.data), oracle_results[name], f"{name}/{mode_name} loaded compiled set")
            got = np.full(w.data.size, -7, dtype=np.int32)
            h2.matchFromHost(w.data.ctypes.data, w.data.size, got.ctypes.data)
            assert_same(got, oracle_results[name], f"{name}/{mode_name} loaded compiled set, matchFromHost")
            h2.destroy()
        # saved without a device, loaded with one
        ho = api.PFAC.createHostOnly()
        ho.setPerfMode(api.PFAC_SPACE_DRIVEN)
        ho.readPatternFromFile(w.pattern_file)
        f2 = str(tmp_path / f"{name}_hostonly.pfacx")
        ho.saveCompiled(f2)
        ho.destroy()
        h3 = api.PFAC.create()
        h3.setKernelVariant(api.PFACX_KERNEL_FILTER)
        h3.loadCompiled(f2)
        assert_same(device_match(h3, w.data), oracle_results[name], f"{name} host-only compiled set on the GPU")
        h3.destroy()


def test_duplicate_patterns_on_the_gpu(tmp_path):
    """Duplicate lines in the pattern file (tests/test_host_api.py::test_duplicate_patterns...) through the kernels."""
    from oracle import binding as ob
    pats = [b"AB", b"CD", b"AB", b"ABX", b"CD", b"Q", b"CDE", b"Q", b"AB"]
    unique = [b"\x01\x02", b"\x01\x03", b"\x01\x04", b"ABX", b"CD", b"\x01\x05", b"CDE", b"Q", b"AB"]
    data = np.frombuffer(b"xxABXyCDEzQABABXCDCDQ" * 3000 + b"AB", dtype=np.uint8)
    fa, fb = tmp_path / "dup.pat", tmp_path / "uniq.pat"
    fa.write_bytes(b"".join(p + b"\n" for p in pats))
    fb.write_bytes(b"".join(p + b"\n" for p in unique))
    want = ob.Oracle(str(fb)).match(data)
    for perf, tex, mode_name in MODES:
        for variant in (api.PFACX_KERNEL_FILTER, api.PFACX_KERNEL_NAIVE):
            h = make_handle(str(fa), perf, tex, variant)
            try:
                assert_same(device_match(h, data), want, f"duplicates / {mode_name} / variant {variant}")
            finally:
                h.destroy()


def test_unified_address_space_buffers_outside_device_memory(workloads, oracle_results):
    """PFAC/test/UVA.cpp: the context lives on GPU 0 while d_input_string / d_matched_result are allocated somewhere
    else in the unified virtual address space (there: a peer GPU).  On a one-GPU box the "somewhere else" is pinned
    host memory: PFAC_matchFromDevice on host-resident buffers must still give the oracle's result -- input streamed
    over the link, zero-fill by the writer waves and patches by the scanning waves ordered on memory they do not own."""
    w = workloads["c3"]
    n = int(w.data.size)
    h_in = torch.from_numpy(w.data.copy()).pin_memory()
    for variant in (api.PFACX_KERNEL_FILTER, api.PFACX_KERNEL_NAIVE):
        for perf, tex, mode_name in (MODES[1], MODES[3]):
            h = make_handle(w.pattern_file, perf, tex, variant)
            try:
                h_out = torch.full((n,), -5, dtype=torch.int32).pin_memory()
                h.matchFromDevice(h_in.data_ptr(), n, h_out.data_ptr())
                torch.cuda.synchronize()
                assert_same(h_out.numpy(), oracle_results["c3"], f"pinned host buffers / {mode_name} / variant {variant}")
                # mixed: input in device memory, result in host memory, and the other way round
                d_in = torch.from_numpy(w.data).to("cuda:0")
                h_out.fill_(-5)
                h.matchFromDevice(d_in.data_ptr(), n, h_out.data_ptr())
                torch.cuda.synchronize()
                assert_same(h_out.numpy(), oracle_results["c3"], f"device input, host result / {mode_name} / variant {variant}")
                d_out = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
                h.matchFromDevice(h_in.data_ptr(), n, d_out.data_ptr())
                torch.cuda.synchronize()
                assert_same(d_out.cpu().numpy(), oracle_results["c3"], f"host input, device result / {mode_name} / variant {variant}")
            finally:
                h.destroy()


def test_kernel_timing_reports_the_filter_kernel_alone(workdir):
    """PFACX_setKernelTiming: HIP events around the filter kernel's launch; PFACX_getScanStats reports its time
    (bench.py: reduce_api.kernel_ms).  Off by default; results do not change."""
    cfg = wl.make_config("c3")
    pf = wl.write_pattern_file(os.path.join(workdir, "timing.pat"), cfg.patterns)
    n = 8 << 20
    d_in = torch.from_numpy(cfg.input_slice(n, 0)).to("cuda:0")
    d_res = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    d_pos = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_AUTOMATIC, api.PFACX_KERNEL_FILTER)      # AUTO would take the tiled kernel for 8 MiB
    try:
        _, count0 = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_res.data_ptr(), d_pos.data_ptr())
        assert "filterKernelMs" not in h.scanStats()
        first = (d_res[:count0].clone(), d_pos[:count0].clone())
        h.setKernelTiming(True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _, count = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_res.data_ptr(), d_pos.data_ptr())
        b.record()
        torch.cuda.synchronize()
        ms = h.scanStats()["filterKernelMs"]
        assert count == count0 and torch.equal(d_res[:count], first[0]) and torch.equal(d_pos[:count], first[1])
        assert 0.0 < ms < a.elapsed_time(b), (ms, a.elapsed_time(b))
        h.matchFromDevice(d_in.data_ptr(), n, d_res.data_ptr())              # the full-result launch is bracketed too
        ms_full = h.scanStats()["filterKernelMs"]
        assert ms_full > 0.0                                # a time was measured; how long it may be is a perf assertion
        if perf_asserts():
            assert ms_full < 5.0, ms_full
        h.setKernelTiming(False)
        h.matchFromDevice(d_in.data_ptr(), n, d_res.data_ptr())
        assert "filterKernelMs" not in h.scanStats()
    finally:
        h.destroy()

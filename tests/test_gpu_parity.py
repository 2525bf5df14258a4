"""GPU parity tests: HIP match path (through the C ABI) == oracle, bit for bit.

Modelled on the reference's example programs: PFAC/test/simple_example.cpp
(matchFromHost), README.md example 2 (matchFromDevice) and the only
self-checking reference test, PFAC/test/omp_PFAC.cpp:396-439 (sliced run ==
single run).  Device memory comes from torch; the match itself never does.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from pfac_amd import api  # noqa: E402

MODES = [
    (api.PFAC_TIME_DRIVEN, api.PFAC_TEXTURE_OFF, "dense-global"),
    (api.PFAC_TIME_DRIVEN, api.PFAC_TEXTURE_ON, "dense-buffer"),
    (api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_OFF, "hash-global"),
    (api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_ON, "hash-buffer"),
]
STAGE = api.PFACX_WALKER_STAGE << 8             # make_handle: the walker of the full-result filter kernel rides in the variant's second byte
VARIANTS = [(api.PFACX_KERNEL_FILTER, "filter"), (api.PFACX_KERNEL_FILTER | STAGE, "filter-stage"), (api.PFACX_KERNEL_NAIVE, "naive"),
            (api.PFACX_KERNEL_AUTO, "auto"), (api.PFACX_KERNEL_REFTABLE, "reftable")]


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available(), "GPU tests need a device; there is no CPU fallback to test"
    torch.cuda.set_device(0)


def make_handle(pattern_file, perf, tex, variant=api.PFACX_KERNEL_FILTER):
    h = api.PFAC.create()
    h.setPerfMode(perf)
    h.setTextureMode(tex)
    h.setKernelVariant(variant & 0xFF)
    if variant >> 8:
        h.setWalker(variant >> 8)                  # (a whole session under one walker: PFAC_TEST_WALKER, pfac_amd/api.py)
    h.readPatternFromFile(pattern_file)
    return h


def device_match(h, data, in_offset=0, out_offset=0):
    """matchFromDevice with poisoned output; optional byte/int offsets to misalign the pointers."""
    n = int(data.size)
    d_in = torch.zeros(n + in_offset + 64, dtype=torch.uint8, device="cuda:0")
    d_in[in_offset:in_offset + n] = torch.from_numpy(np.ascontiguousarray(data)).to("cuda:0")
    d_out = torch.full((n + out_offset + 64,), -5, dtype=torch.int32, device="cuda:0")
    h.matchFromDevice(d_in.data_ptr() + in_offset, n, d_out.data_ptr() + 4 * out_offset)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    assert np.all(out[:out_offset] == -5) and np.all(out[out_offset + n:] == -5), "wrote outside [0, n)"
    return out[out_offset:out_offset + n]


def assert_same(got, want, what):
    if not np.array_equal(got, want):
        bad = np.nonzero(got != want)[0]
        raise AssertionError(f"{what}: {bad.size} mismatches; first at {bad[0]}: got {got[bad[0]]} want {want[bad[0]]}")


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


def test_match_from_host_pipelined_pieces_equal_match_from_device(workdir):
    """PFAC_matchFromHost on the GPU platform scans in 32 Mi-position pieces with overlapped copies
    (SURVEY 8f rank 2): a 70 MiB stream (three pieces, the last one ragged) must give what one
    PFAC_matchFromDevice call over the whole stream gives, including matches that straddle the cuts."""
    import os
    from pfac_amd import workloads as wl
    pats = wl.snort_patterns(2000)
    pf = wl.write_pattern_file(os.path.join(workdir, "hostpipe.pat"), pats)
    n = (70 << 20) + 12345
    data = wl.http_stream(n, wl.http_message_pool(pats, pool_size=256, embed_fraction=0.3)).copy()
    longest = max(pats, key=len)
    straddlers = [((32 << 20) - 1, longest), ((64 << 20) - len(longest) // 2, longest), ((64 << 20) - 300, longest)]
    for at, p in straddlers:
        data[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_OFF)
    try:
        want = device_match(h, data)
        got = np.full(n, -7, dtype=np.int32)
        h.matchFromHost(data.ctypes.data, n, got.ctypes.data)
        assert_same(got, want, "pipelined matchFromHost")
        assert all(want[at] != 0 for at, _ in straddlers)                        # matches that straddle a cut
        got2 = np.full(1000, -7, dtype=np.int32)                                   # a call smaller than the staging buffers
        h.matchFromHost(data.ctypes.data, 1000, got2.ctypes.data)
        assert_same(got2[:900], want[:900], "small matchFromHost after a large one")
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


@pytest.mark.parametrize("workload,perf", [("c3", api.PFAC_SPACE_DRIVEN), ("c2", api.PFAC_TIME_DRIVEN)])
def test_full_size_properties(workdir, workload, perf):
    """BASELINE.json sizes (1 GiB): size-independent properties instead of a full oracle pass.
       (a) two independent kernels (prefilter+walkers vs one-thread-per-byte) agree on every element;
       (b) scanning the stream in 5 slices with maxPatternLen+1 overlap reproduces the single call
           (the reference's omp_PFAC.cpp self-check);
       (c) sampled 256 KiB windows equal the oracle;  (d) every planted pattern is reported."""
    from oracle import binding as ob
    from pfac_amd import sharding
    from pfac_amd import workloads as wl
    n = 1 << 30
    cfg = wl.make_config(workload)
    pf = wl.write_pattern_file(f"{workdir}/full_{workload}.pat", cfg.patterns)
    host = cfg.input_slice(n, 0)
    rng = np.random.Generator(np.random.PCG64(17))
    planted = []
    for _ in range(64):
        pid = int(rng.integers(0, len(cfg.patterns)))
        at = int(rng.integers(0, n - 128))
        p = np.frombuffer(cfg.patterns[pid], dtype=np.uint8)
        host[at:at + p.size] = p
        planted.append((at, pid + 1, p.size))
    edge = np.frombuffer(cfg.patterns[3], dtype=np.uint8)
    host[n - edge.size:] = edge                      # a match that ends exactly at the last byte
    d_in = torch.from_numpy(host).to("cuda:0")
    d_a = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
    h = make_handle(pf, perf, api.PFAC_AUTOMATIC)
    try:
        h.matchFromDevice(d_in.data_ptr(), n, d_a.data_ptr())
        torch.cuda.synchronize()
        assert int((d_a < 0).sum()) == 0, "every element must be written"
        # (a)
        d_b = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
        h.setKernelVariant(api.PFACX_KERNEL_NAIVE)
        h.matchFromDevice(d_in.data_ptr(), n, d_b.data_ptr())
        h.setKernelVariant(api.PFACX_KERNEL_FILTER)
        torch.cuda.synchronize()
        assert torch.equal(d_a, d_b), "filter kernel and naive kernel disagree"
        # (b)
        d_b.fill_(-1)
        overlap = sharding.overlap_bytes(h.info().maxPatternLen)
        scratch = torch.empty((n // 5 + overlap + 4096,), dtype=torch.int32, device="cuda:0")
        for s in sharding.plan_slices(n, 5, overlap):
            h.matchFromDevice(d_in.data_ptr() + s.start, s.read_end - s.start, scratch.data_ptr())
            d_b[s.start:s.end] = scratch[: s.end - s.start]
        torch.cuda.synchronize()
        assert torch.equal(d_a, d_b), "sliced scan differs from the single call"
        del d_b, scratch
        # (d)
        res_at = d_a[torch.tensor([p[0] for p in planted], device="cuda:0")].cpu().numpy()
        assert np.all(res_at != 0)
        lens = np.array([0] + [len(p) for p in cfg.patterns])
        assert np.all(lens[res_at] >= np.array([p[2] for p in planted])), "longest-match semantics"
        assert int(d_a[n - edge.size]) != 0
        # (c)
        o = ob.Oracle(pf, dense=(perf == api.PFAC_TIME_DRIVEN), hashed=(perf == api.PFAC_SPACE_DRIVEN))
        win, tail = 1 << 18, o.max_pattern_len + 1
        for s in [0, n - win] + [int(x) for x in rng.integers(0, n - win, size=6)]:
            want = o.match(host[s:min(n, s + win + tail)], hashed=(perf == api.PFAC_SPACE_DRIVEN), omp=True)[:win]
            assert_same(d_a[s:s + win].cpu().numpy(), want, f"window at {s}")
        o.close()
    finally:
        h.destroy()


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


def test_input_larger_than_4_gib(workdir):
    """The vector kernel keeps positions in 32 bits; inputs of 4 GiB and more are scanned as
    consecutive windows (scan_module.hip: kMaxLaunchBytes) whose overlap is rewritten by the next
    window.  Patterns planted across the window boundary, across 2^32 and at the very end must be
    reported, and the whole vector must equal those of the tiled kernel (64-bit group offsets) and of the reference-shaped
    kernel (size_t positions, reference-layout table).
    ANCHOR: the reference's sizes are `int` (< 2 GiB), so there is no reference output at this size; this is three
    independent kernels against each other plus the planted patterns, whose expected IDs come from the pattern file.
    The same stream's first 1 GiB is pinned on the reference digest in test_gpu_round2.py."""
    from pfac_amd import workloads as wl
    cfg = wl.make_config("c2")
    pf = wl.write_pattern_file(f"{workdir}/big.pat", cfg.patterns)
    n = (1 << 32) + (3 << 20) + 5
    window = (1 << 32) - (1 << 24)
    d_in = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    chunk = cfg.input_slice(1 << 28, 0)
    t = torch.from_numpy(chunk).to("cuda:0")
    for off in range(0, n, 1 << 28):
        m = min(1 << 28, n - off)
        d_in[off:off + m] = t[:m]
    del t
    planted = []
    lens = [len(p) for p in cfg.patterns]
    longest = int(np.argmax(lens))
    for at, pid in [(window - 16, longest), (window - 100, 3), (window + 64, 4), ((1 << 32) - 7, longest),
                    ((1 << 32) + 200, 9), (n - lens[11], 11), (12345, 12), (window - 200 - lens[5], 5)]:
        p = torch.tensor(list(cfg.patterns[pid]), dtype=torch.uint8, device="cuda:0")
        d_in[at:at + p.numel()] = p
        planted.append((at, pid + 1))
    h = make_handle(pf, api.PFAC_TIME_DRIVEN, api.PFAC_AUTOMATIC)
    try:
        d_a = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
        h.matchFromDevice(d_in.data_ptr(), n, d_a.data_ptr())
        torch.cuda.synchronize()
        for at, pid in planted:
            assert int(d_a[at]) == pid, f"planted pattern {pid} at {at}: got {int(d_a[at])}"
        d_b = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
        for variant in (api.PFACX_KERNEL_NAIVE, api.PFACX_KERNEL_REFTABLE):
            d_b.fill_(-1)
            h.setKernelVariant(variant)
            h.matchFromDevice(d_in.data_ptr(), n, d_b.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(d_a, d_b), f"variant {variant}"
    finally:
        h.destroy()


@pytest.mark.parametrize("perf,tex,mode_name", MODES)
@pytest.mark.parametrize("name", ["c1", "ex2", "c2", "c3", "c5", "dense_hits", "binary"])
def test_match_from_device_reduce_equals_oracle(workloads, oracle_results, name, perf, tex, mode_name):
    """PFAC_matchFromDeviceReduce: compacted (id, position) pairs in ascending position order
    (ref PFAC.cpp:964-1008; known answer user guide r1.2 p.29 is the c1 case)."""
    from oracle import binding as ob
    w = workloads[name]
    ids, pos = ob.reduce(oracle_results[name])
    n = int(w.data.size)
    h = make_handle(w.pattern_file, perf, tex)
    try:
        d_in = torch.from_numpy(w.data.copy()).to("cuda:0")
        d_res = torch.full((n + 8,), -5, dtype=torch.int32, device="cuda:0")
        d_pos = torch.full((n + 8,), -5, dtype=torch.int32, device="cuda:0")
        st, count = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_res.data_ptr(), d_pos.data_ptr())
        torch.cuda.synchronize()
        assert count == ids.size, f"{name}/{mode_name}: count {count} != {ids.size}"
        assert np.array_equal(d_pos[:count].cpu().numpy(), pos), f"{name}/{mode_name} positions"
        assert np.array_equal(d_res[:count].cpu().numpy(), ids), f"{name}/{mode_name} ids"
        assert int(d_res[n:].min()) == -5 and int(d_pos[n:].min()) == -5, "wrote past the caller's arrays"
    finally:
        h.destroy()


def test_match_from_host_reduce_on_gpu_platform(workloads, oracle_results, golden_dir):
    """PFAC_matchFromHostReduce on PFAC_PLATFORM_GPU (ref PFAC.cpp:1010-1128, simple_example_reduce.cpp)."""
    import json, os
    from oracle import binding as ob
    ka = json.load(open(os.path.join(golden_dir, "known_answers.json")))["example1"]
    for name in ("c1", "c3"):
        w = workloads[name]
        ids, pos = ob.reduce(oracle_results[name])
        h = make_handle(w.pattern_file, api.PFAC_SPACE_DRIVEN, api.PFAC_AUTOMATIC)
        try:
            res = np.full(w.data.size, -5, dtype=np.int32)
            hp = np.full(w.data.size, -5, dtype=np.int32)
            st, count = h.matchFromHostReduce(w.data.ctypes.data, w.data.size, res.ctypes.data, hp.ctypes.data)
            assert count == ids.size and np.array_equal(res[:count], ids) and np.array_equal(hp[:count], pos)
            if name == "c1":
                assert hp[:count].tolist() == ka["reduce_pos"] and res[:count].tolist() == ka["reduce_id"]
        finally:
            h.destroy()


def test_reduce_misaligned_and_tiny_inputs(workloads, oracle_results):
    from oracle import binding as ob
    w = workloads["dense_hits"]
    for n, off in [(5, 0), (1000, 0), (2049, 1), (30000, 3)]:
        data = np.tile(w.data, n // w.data.size + 2)[7:7 + n].copy()
        o = ob.Oracle(w.pattern_file, hashed=False)
        ids, pos = ob.reduce(o.match(data))
        o.close()
        h = make_handle(w.pattern_file, api.PFAC_TIME_DRIVEN, api.PFAC_TEXTURE_OFF)
        try:
            d_in = torch.zeros(n + 64, dtype=torch.uint8, device="cuda:0")
            d_in[off:off + n] = torch.from_numpy(data).to("cuda:0")
            d_res = torch.zeros(n + 8, dtype=torch.int32, device="cuda:0")
            d_pos = torch.zeros(n + 8, dtype=torch.int32, device="cuda:0")
            st, count = h.matchFromDeviceReduce(d_in.data_ptr() + off, n, d_res.data_ptr(), d_pos.data_ptr())
            assert count == ids.size
            assert np.array_equal(d_pos[:count].cpu().numpy(), pos) and np.array_equal(d_res[:count].cpu().numpy(), ids)
        finally:
            h.destroy()

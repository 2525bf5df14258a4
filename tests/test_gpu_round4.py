"""GPU parity tests, round 4: the tiled kernel (pfac_scan_tiled: calls below 32 MiB, PFACX_KERNEL_NAIVE, the filter kernel's
pattern-dense chunks) in both of its shapes, both of its modes (compacted survivors / dense groups walked in place) and
both outputs; the reference-layout tables that now only exist on request.

Reference model: the result contract of PFAC/src/PFAC_kernel.cu:377-458 (d_matched_result[j] = ID of the longest pattern
starting at byte j, else 0, every element written) against the oracle, bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from pfac_amd import api  # noqa: E402
from pfac_amd import workloads as wl  # noqa: E402
from tests.test_gpu_parity import MODES, assert_same, device_match, make_handle  # noqa: E402


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

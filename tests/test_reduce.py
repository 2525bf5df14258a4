"""PFAC_matchFromDeviceReduce / PFAC_matchFromHostReduce (compacted output: ids and positions in ascending position order) against the
oracle.  Reference: PFAC/src/PFAC_reduce_kernel.cu:172-295, PFAC_reduce_inplace_kernel.cu:155-323, user guide r1.2 p.29."""

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


@pytest.mark.parametrize("n,everywhere", [((3 << 20) + 77, False), ((48 << 20) + 5, False), ((300 << 20) + 1, False), ((1 << 30) + 4097, False),
                                          ((3 << 29) + 123, False), ((4 << 20) + 9, True), ((96 << 20) + 1, True)])
def test_compacted_output_is_in_position_order_at_every_bin_shape(workdir, n, everywhere):
    """PFAC_matchFromDeviceReduce orders its pairs with position bins (scan_order.inc: PairOrder): the bin width follows
    the input size (64 positions ... 32 Ki positions), a bin with more than 64 pairs is ranked through a bitmap in LDS.
    Inputs with crowded stretches (one position in eight matches) between sparse ones, at sizes that take every bin
    width class -- and crowded everywhere: more pairs than the handle's scratch holds on a first call, the launches leave
    and are queued again behind a larger one.
    ANCHOR: HIP-vs-HIP at these sizes -- expected = the non-zero entries of the FULL result of the same handle, which is the
    reference's definition of the compacted output (PFAC_reduce_kernel.cu:417-457: a stable compaction of the full result);
    the full-result path is what tests/test_full_result.py / test_full_size_digests.py pin on the oracle and the reference digests, and the
    patterns here are planted ones whose matches can be counted by hand (one per 'h', ...)."""
    pats = [b"h", b"ab", b"abc", b"gfe", b"mnop", b"xyzzy", b"qq", b"nopqrstu"]
    pf = wl.write_pattern_file(os.path.join(workdir, f"order{n}.pat"), pats)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(n & 0xFFFF)
    crowded = torch.randint(97, 105, (n,), dtype=torch.uint8, device="cuda:0", generator=g)       # 'a'..'h'
    d_in = torch.randint(105, 123, (n,), dtype=torch.uint8, device="cuda:0", generator=g)         # 'i'..'z'
    for lo, hi in (((0, n),) if everywhere else ((0, 70_000), (n // 3, n // 3 + (n >> 5)), (n - 50_000, n))):   # start, a stretch inside, the very end
        d_in[lo:hi] = crowded[lo:hi]
    del crowded
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_AUTOMATIC, api.PFACX_KERNEL_AUTO)
    try:
        d_full = torch.empty(n, dtype=torch.int32, device="cuda:0")
        h.matchFromDevice(d_in.data_ptr(), n, d_full.data_ptr())
        want_pos = torch.nonzero(d_full).flatten()
        want_ids = d_full[want_pos]
        del d_full
        d_res = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
        d_pos = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
        for _ in range(2):                                       # the second call finds the scratch of the first
            st, count = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_res.data_ptr(), d_pos.data_ptr())
            torch.cuda.synchronize()
            assert count == want_pos.numel() and count > n >> 9
            assert torch.equal(d_pos[:count].to(torch.int64), want_pos), "positions"
            assert torch.equal(d_res[:count], want_ids), "pattern IDs"
            assert int(d_pos[count:].max()) == -5 and int(d_res[count:].max()) == -5, "wrote behind the pairs"
            d_res.fill_(-5)
            d_pos.fill_(-5)
    finally:
        h.destroy()


def test_compacted_output_calls_leave_their_counters_clean(workdir):
    """PFAC_matchFromDeviceReduce keeps state between calls since round 5 (scan_order.inc, scan_module.hip: reduceScan): the ordering
    launches leave the bin counters zero and the next call skips its memset when it finds the same layout; the pairs' counter
    exists twice and calls alternate; the count comes back through mapped host memory.  Sequences that change everything that
    state depends on -- the input size (another bin layout), a call whose pairs do not fit the scratch (second round through the
    four kernels behind a larger scratch), PFAC_matchFromHost in between (its pieces run the unordered path on the same scratch),
    PFACX_trim (the scratch is gone), the tiled kernel (small inputs) -- and after each step the pairs must be exactly the non-zero
    entries of the full result in position order.
    ANCHOR: the full result of the same handle (pinned on the oracle / the reference digests by tests/test_full_result.py), as in
    test_compacted_output_is_in_position_order_at_every_bin_shape."""
    pats = [b"h", b"ab", b"abc", b"gfe", b"mnop", b"xyzzy", b"qq", b"nopqrstu"]
    pf = wl.write_pattern_file(os.path.join(workdir, "tidy.pat"), pats)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(77)
    big = (96 << 20) + 13
    sparse = torch.randint(105, 123, (big,), dtype=torch.uint8, device="cuda:0", generator=g)           # 'i'..'z': mnop, xyzzy, qq, nopqrstu
    crowded = torch.randint(97, 105, (40 << 20,), dtype=torch.uint8, device="cuda:0", generator=g)      # 'a'..'h': one position in eight matches
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_AUTOMATIC, api.PFACX_KERNEL_AUTO)
    d_res = torch.empty(big, dtype=torch.int32, device="cuda:0")
    d_pos = torch.empty(big, dtype=torch.int32, device="cuda:0")
    d_full = torch.empty(big, dtype=torch.int32, device="cuda:0")

    def check(d_in, n, what):
        h.matchFromDevice(d_in.data_ptr(), n, d_full.data_ptr())
        want_pos = torch.nonzero(d_full[:n]).flatten()
        want_ids = d_full[:n][want_pos]
        d_res.fill_(-5)
        d_pos.fill_(-5)
        _, count = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_res.data_ptr(), d_pos.data_ptr())
        torch.cuda.synchronize()
        assert count == want_pos.numel(), (what, count, want_pos.numel())
        assert torch.equal(d_pos[:count].to(torch.int64), want_pos) and torch.equal(d_res[:count], want_ids), what
        assert int(d_pos[count:].max()) == -5 and int(d_res[count:].max()) == -5, what
        return count

    try:
        first = check(sparse, big, "first call")
        assert check(sparse, big, "same layout: no memset") == first
        assert check(sparse, big, "same layout again: the other pair of counters") == first
        check(sparse, (33 << 20) + 5, "smaller input: another bin layout")
        assert check(sparse, big, "back to the first layout") == first
        many = check(crowded, 40 << 20, "more pairs than the scratch holds")
        assert many > (40 << 20) // 16
        check(crowded, 40 << 20, "... and again, behind the larger scratch")
        assert check(sparse, big, "sparse again") == first
        host = np.frombuffer(sparse[:(48 << 20) + 3].cpu().numpy().tobytes(), dtype=np.uint8)
        got = np.empty(host.size, dtype=np.int32)
        h.matchFromHost(host.ctypes.data, host.size, got.ctypes.data)                                      # unordered pieces on the same scratch
        h.matchFromDevice(sparse.data_ptr(), host.size, d_full.data_ptr())
        assert np.array_equal(got, d_full[:host.size].cpu().numpy()), "PFAC_matchFromHost"
        assert check(sparse, big, "after PFAC_matchFromHost") == first
        h.trim()
        assert check(sparse, big, "after PFACX_trim") == first
        check(sparse[5:], (1 << 20) + 1, "a small, misaligned input: the tiled kernel")
        check(sparse, 700, "700 bytes")
        assert check(sparse, big, "and the big one once more") == first
    finally:
        h.destroy()


def test_cpp_reduce_example_prints_the_user_guide_answer():
    """examples/reduce_example.cpp (re-authored PFAC/test/simple_example_reduce.cpp): user guide r1.2 p.29 --
    h_num_matched = 5, positions {0,1,2,4,6}, patterns {1,3,4,4,2} -- from matchFromHostReduce in both perf modes
    and from matchFromDeviceReduce."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-B"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "examples", "reduce_example")], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.splitlines() == ["number of matched = 5", "At position    0, match pattern 1", "At position    1, match pattern 3",
                                       "At position    2, match pattern 4", "At position    4, match pattern 4",
                                       "At position    6, match pattern 2"]

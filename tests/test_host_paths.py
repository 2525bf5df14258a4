"""Host buffers through the GPU (the PCIe-inclusive paths): PFAC_matchFromHost and PFAC_matchFromHostReduce as pipelines of pieces, against
the oracle; one handle shared by host threads.  Reference: PFAC/src/PFAC.cpp:879-961, 1010-1128; SimpleMultiGPU_pthread.cpp:50-174."""

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


def test_match_from_host_pieces_equal_oracle(workdir):
    """PFAC_matchFromHost on the GPU platform scans in 32 Mi-position pieces with overlapped copies (SURVEY 8f
    rank 2): a 70 MiB stream (three pieces) and a 64 MiB + 10 stream (last piece shorter than the longest
    pattern) against the ORACLE, including matches that straddle the cuts; matchFromDevice as well."""
    from oracle import binding as ob
    pats = wl.snort_patterns(2000)
    pf = wl.write_pattern_file(os.path.join(workdir, "hostpipe2.pat"), pats)
    n = (70 << 20) + 12345
    data = wl.http_stream(n, wl.http_message_pool(pats, pool_size=256, embed_fraction=0.3)).copy()
    longest = max(pats, key=len)
    straddlers = [(32 << 20) - 1, (64 << 20) - len(longest) // 2, (64 << 20) - 300]
    for at in straddlers:
        data[at:at + len(longest)] = np.frombuffer(longest, dtype=np.uint8)
    o = ob.Oracle(pf, dense=False, hashed=True)
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_OFF)
    try:
        for size in (n, (64 << 20) + 10):
            if size != n:                                        # a match that ends with the 10-byte last piece
                data[size - len(longest):size] = np.frombuffer(longest, dtype=np.uint8)
                straddlers = [(32 << 20) - 1, size - len(longest)]
            want = o.match(data[:size], hashed=True, omp=True)
            got = np.full(size, -7, dtype=np.int32)
            h.matchFromHost(data.ctypes.data, size, got.ctypes.data)
            assert_same(got, want, f"pipelined matchFromHost, {size} bytes")
            assert_same(device_match(h, data[:size]), want, f"matchFromDevice, {size} bytes")
            assert all(want[at] != 0 for at in straddlers)      # matches that straddle a cut
    finally:
        h.destroy()
        o.close()


def test_match_from_host_sparse_and_dense_pieces(workdir):
    """PFAC_matchFromHost brings back compacted (position, id) pairs and fills the zeros on the host; a piece in which
    more than one position in eight matches takes the full-vector route.  A 42 MiB stream whose first piece is
    sparse and whose second piece ends in 8 MiB where every position matches, against the ORACLE; the result vector
    starts out as garbage (every element must be written)."""
    from oracle import binding as ob
    pats = [b"a", b"aa", b"aaa", b"aaaa", b"ab", b"b" * 7, b"abc" * 5] + wl.snort_patterns(500)
    pf = wl.write_pattern_file(os.path.join(workdir, "hostsparse.pat"), pats)
    n = (42 << 20) + 77
    data = wl.http_stream(n, wl.http_message_pool(pats[7:], pool_size=128, embed_fraction=0.3)).copy()
    data[data == ord("a")] = ord("e")                              # keeps the text part sparse: no 1-byte hits
    data[(34 << 20):] = ord("a")                                   # ... and the end of the second piece as dense as it gets
    data[(33 << 20):(33 << 20) + 15] = np.frombuffer(b"abc" * 5, dtype=np.uint8)
    o = ob.Oracle(pf, dense=False, hashed=True)
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_AUTOMATIC)
    try:
        want = o.match(data, hashed=True, omp=True)
        first, second = want[:32 << 20], want[32 << 20:]
        assert 0 < np.count_nonzero(first) < first.size // 8 and np.count_nonzero(second) > second.size // 8
        for trial in range(2):                                     # the second call reuses the staging buffers
            got = np.full(n, -7, dtype=np.int32)
            h.matchFromHost(data.ctypes.data, n, got.ctypes.data)
            assert_same(got, want, f"matchFromHost, sparse + dense pieces, call {trial}")
    finally:
        h.destroy()
        o.close()


def test_match_from_host_reduce_pieces_equal_oracle(workdir):
    """PFAC_matchFromHostReduce on the GPU platform (ref PFAC.cpp:1010-1128; known answer user guide r1.2 p.29) goes through the
    same staged pieces as PFAC_matchFromHost -- 16 Mi positions each, piece i + 1 uploading while piece i is scanned -- and the
    pairs come back in position order across the pieces.  A 70 MiB text stream (five pieces) with the longest pattern planted
    across every cut, and a 42 MiB stream whose end matches at every position (3.4 M pairs from one piece), against the
    oracle's result vector; the device memory the call leaves allocated is two pieces, not 9 bytes for every position."""
    from oracle import binding as ob
    pats = [b"a", b"aa", b"aaa", b"aaaa", b"ab", b"b" * 7, b"abc" * 5] + wl.snort_patterns(2000)
    pf = wl.write_pattern_file(os.path.join(workdir, "hostreduce.pat"), pats)
    longest = max(pats, key=len)
    piece = 16 << 20
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_ON, api.PFACX_KERNEL_AUTO)
    o = ob.Oracle(pf, dense=False, hashed=True)
    try:
        for which, n in (("text", (70 << 20) + 12345), ("dense end", (42 << 20) + 77), ("one small piece", 100003)):
            data = wl.http_stream(n, wl.http_message_pool(pats[7:], pool_size=256, embed_fraction=0.3)).copy()
            data[data == ord("a")] = ord("e")
            if which == "dense end":
                data[(39 << 20):] = ord("a")
            for k in range(1, n // piece + 1):
                for at in (k * piece - 1, k * piece - len(longest) // 2, k * piece - len(longest)):
                    if 0 <= at and at + len(longest) <= n:
                        data[at:at + len(longest)] = np.frombuffer(longest, dtype=np.uint8)
            want = o.match(data, hashed=True, omp=True)
            nz = np.flatnonzero(want)
            ids = np.full(n, -9, dtype=np.int32)
            pos = np.full(n, -9, dtype=np.int32)
            for trial in range(2):
                _, count = h.matchFromHostReduce(data.ctypes.data, n, ids.ctypes.data, pos.ctypes.data)
                assert count == nz.size, (which, count, nz.size)
                assert np.array_equal(pos[:count], nz) and np.array_equal(ids[:count], want[nz]), which
            info = h.info()
            # two staged pieces (9 bytes per position) + the scratch the pairs of one piece are ordered through (grows with the densest piece seen)
            assert info.deviceTableBytes + info.deviceScratchBytes <= (300 << 20 if which == "text" else 400 << 20), (which, info.deviceTableBytes, info.deviceScratchBytes)
    finally:
        h.destroy()
        o.close()



def test_two_host_threads_share_one_handle(workloads, oracle_results):
    """The reference serialises threads that share a handle with its texture mutex (PFAC.cpp:37-56).  Here two
    host threads call matchFromDevice, matchFromDeviceReduce and matchFromHost on ONE handle at the same time:
    every call must return its own complete, correct result (chunk counters, match counter, sort scratch and
    staging buffers are per-handle state)."""
    w = workloads["c3"]
    want = oracle_results["c3"]
    nz = np.flatnonzero(want)
    h = make_handle(w.pattern_file, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_ON)
    n = int(w.data.size)
    errors = []

    def worker(k):
        try:
            torch.cuda.set_device(0)
            d_in = torch.from_numpy(w.data).to("cuda:0")
            for it in range(12):
                d_out = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
                d_pos = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
                if (it + k) % 3 == 0:
                    h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
                    torch.cuda.synchronize()
                    assert_same(d_out.cpu().numpy(), want, f"thread {k} call {it} matchFromDevice")
                elif (it + k) % 3 == 1:
                    _, count = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_out.data_ptr(), d_pos.data_ptr())
                    assert count == nz.size and np.array_equal(d_pos[:count].cpu().numpy(), nz) and \
                        np.array_equal(d_out[:count].cpu().numpy(), want[nz]), f"thread {k} call {it} reduce"
                else:
                    got = np.full(n, -7, dtype=np.int32)
                    h.matchFromHost(w.data.ctypes.data, n, got.ctypes.data)
                    assert_same(got, want, f"thread {k} call {it} matchFromHost")
        except Exception as e:                                   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    h.destroy()
    assert not errors, errors[0]


def test_match_from_host_pinned_keeps_up_with_pageable(workdir):
    """PFAC_matchFromHost from pinned buffers must not fall behind the pageable path (round 4's driver line: 29 against 49 GB/s as
    medians on a two-socket host; the fill threads now run on the NUMA node of the caller's result vector).  Medians of 12
    interleaved calls on 128 MiB of the Snort-style stream, results equal; the bound is loose (0.8) because single calls do stall on shared hosts."""
    import time
    cfg = wl.make_config("c3")
    pf = wl.write_pattern_file(os.path.join(workdir, "pinned.pat"), cfg.patterns)
    n = 128 << 20
    host = cfg.input_slice(n, 0).copy()
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_ON, api.PFACX_KERNEL_AUTO)
    try:
        bufs = {}
        for kind in ("pageable", "pinned"):
            h_in, h_out = torch.from_numpy(host.copy()), torch.full((n,), -7, dtype=torch.int32)
            if kind == "pinned":
                h_in, h_out = h_in.pin_memory(), h_out.pin_memory()
            bufs[kind] = (h_in, h_out)
            for _ in range(2):
                h.matchFromHost(h_in.data_ptr(), n, h_out.data_ptr())
        assert np.array_equal(bufs["pinned"][1].numpy(), bufs["pageable"][1].numpy()) and np.count_nonzero(bufs["pinned"][1].numpy()) > 1000
        if not perf_asserts():                                 # a correctness run ends here: the rest compares wall-clock times (gpu_helpers.perf_asserts)
            return
        # the two kinds take turns (a stall of the shared host hits both), medians of 12; a shared host can still stall one side of a
        # whole attempt: three attempts, one has to hold
        seen, best = [], {"pageable": float("inf"), "pinned": float("inf")}
        for attempt in range(3):
            ts = {"pageable": [], "pinned": []}
            for _ in range(12):
                for kind in ("pageable", "pinned"):
                    h_in, h_out = bufs[kind]
                    t0 = time.perf_counter()
                    h.matchFromHost(h_in.data_ptr(), n, h_out.data_ptr())
                    ts[kind].append(time.perf_counter() - t0)
            med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
            best = {k: min(best[k], min(v)) for k, v in ts.items()}
            seen.append(med)
            if n / med["pinned"] >= 0.8 * (n / med["pageable"]):
                break
        else:
            # a host whose memory channels are busy for the whole test: what the path CAN do (best calls of all attempts) still has to hold
            assert n / best["pinned"] >= 0.8 * (n / best["pageable"]), (seen, best)
    finally:
        h.destroy()

"""BASELINE.json sizes: the whole result vector of a 1 GiB (8 GiB, 4.3 GiB) call against the digests of the REFERENCE's own PFAC_CPU_OMP
output (tests/golden/full_digests.json, made by tests/golden/make_full_digests.py from oracle/_ref) and through size-independent
properties.  Reference model: PFAC/test/omp_PFAC.cpp:257-439 (sliced run == single run)."""

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


@pytest.mark.parametrize("workload,perf,slice_index,variant", [
    ("c2", api.PFAC_TIME_DRIVEN, 0, "last"), ("c3", api.PFAC_SPACE_DRIVEN, 0, "last"),
    ("c5", api.PFAC_TIME_DRIVEN, 0, "last"), ("c5", api.PFAC_SPACE_DRIVEN, 0, "last"),
    ("c3", api.PFAC_SPACE_DRIVEN, 1, "inner"),
])
def test_full_size_result_equals_reference_digest(workdir, workload, perf, slice_index, variant):
    """BASELINE.json sizes (1 GiB): the whole int32 result vector of PFAC_matchFromDevice has the SHA-256,
    FNV-1a-64, match count and position checksum of the REFERENCE's own PFAC_CPU_OMP result on the same stream
    (tests/golden/full_digests.json, produced by tests/golden/make_full_digests.py from oracle/_ref).  `inner`
    = a slice of the multi-GPU stream scanned with the head of its successor (BASELINE config 4)."""
    n = 1 << 30
    rec = digest_record(workload, slice_index, 1024)
    cfg = wl.make_config(workload)
    pf = wl.write_pattern_file(f"{workdir}/digest_{workload}.pat", cfg.patterns)
    assert wl.fnv1a(np.fromfile(pf, dtype=np.uint8)) == rec["pattern_file_fnv1a"], "pattern generator drifted"
    overlap = rec["overlap"] if variant == "inner" else 0
    host = np.empty(n + overlap, dtype=np.uint8)
    host[:n] = cfg.input_slice(n, slice_index)
    if overlap:
        host[n:] = cfg.input_slice(overlap, slice_index + 1)
    assert wl.fnv1a(host[:n]) == rec["input_fnv1a"], "input generator drifted"
    d_in = torch.from_numpy(host).to("cuda:0")
    d_out = torch.full((n + overlap,), -1, dtype=torch.int32, device="cuda:0")
    h = make_handle(pf, perf, api.PFAC_AUTOMATIC)
    try:
        h.matchFromDevice(d_in.data_ptr(), n + overlap, d_out.data_ptr())
        torch.cuda.synchronize()
    finally:
        h.destroy()
    got = d_out[:n].cpu().numpy()
    del d_in, d_out
    want = rec[variant]
    pos = np.flatnonzero(got)
    assert int(pos.size) == want["match_count"]
    assert sharding.position_checksum(pos, got[pos], base=slice_index * n) == want["checksum"]
    assert wl.fnv1a_sparse_i32(pos, got[pos], n) == want["fnv1a64"]
    assert hashlib.sha256(got.view(np.uint8)).hexdigest() == want["sha256"]


@pytest.mark.parametrize("workload,perf,walker,walker_name", [("c5", api.PFAC_TIME_DRIVEN, w, name) for w, name in WALKERS] +
                         [("c6", api.PFAC_SPACE_DRIVEN, w, name) for w, name in WALKERS + [(api.PFACX_WALKER_VETO, "veto"), (api.PFACX_WALKER_AUTO, "auto")]])
def test_full_size_near_miss_stream_equals_reference_digest(workdir, workload, perf, walker, walker_name):
    """The near-miss stream at full size (1 GiB) under every walker: count, checksum and FNV-1a-64 of the result vector equal the
    reference's PFAC_CPU_OMP digest (tests/golden/full_digests.json); steps per walk as round 5's target asks (<= 4.5).
    c5 = BASELINE config 5 (2 000 patterns: its tail table lies in LDS, WINDOW = the veto kernel with VETO = 1); c6 = the same stream
    over C3's 30 000 Snort-style patterns + C5's 1 000 shared-prefix patterns (PFAC_hash_draft.pdf Table 5: the worst input on the
    FULL set): the table lies in device memory, PFACX_WALKER_VETO = the VETO = 2 kernel, and PFACX_WALKER_AUTO gets there by itself
    with its third launch (the first runs the plain window walker and reports near misses, the second was queued before that)."""
    ref = digest_record(workload, 0, 1024)["last"]
    cfg = wl.make_config(workload)
    pf = wl.write_pattern_file(os.path.join(workdir, workload + "full.pat"), cfg.patterns)
    n = 1 << 30
    d_in = torch.from_numpy(cfg.input_slice(n, 0)).to("cuda:0")
    d_out = torch.empty(n, dtype=torch.int32, device="cuda:0")
    h = make_handle(pf, perf, api.PFAC_TEXTURE_ON, api.PFACX_KERNEL_AUTO | (walker << 8))
    if walker == api.PFACX_WALKER_AUTO:
        h.setWalker(api.PFACX_WALKER_AUTO)          # (a session under PFAC_TEST_WALKER forces every new handle's walker)
    try:
        for _ in range(2):
            h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
            torch.cuda.synchronize()
        if walker == api.PFACX_WALKER_AUTO:
            d_out.fill_(-3)
            h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
            torch.cuda.synchronize()
        st = h.scanStats()
        info = h.info()
        if walker == api.PFACX_WALKER_AUTO:
            assert st["veto"] == 2 and info.filterTailGlobalEntries > 0, (st, info.filterTailGlobalEntries)
        elif walker == api.PFACX_WALKER_VETO:
            assert st["veto"] == 2 and st["walker"] == api.PFACX_WALKER_WINDOW, st
        else:
            assert st["walker"] == walker
            assert st["veto"] == (1 if (walker == api.PFACX_WALKER_WINDOW and info.filterTailEntries > 0) else 0), st
        out = d_out.cpu().numpy()
        pos = np.flatnonzero(out)
        assert int(pos.size) == ref["match_count"]
        assert wl.fnv1a_sparse_i32(pos, out[pos], n) == ref["fnv1a64"]
        steps = st["laneSteps"] / max(st["walksStarted"], 1)
        assert steps <= (4.5 if walker == api.PFACX_WALKER_STAGE else 5.0), steps
        if st["veto"]:                                  # the veto spares the walks: a quarter of the stream's 22 M candidates per GiB at most
            assert st["walksStarted"] < 6.0e6, st
    finally:
        h.destroy()


def test_all_eight_slices_of_the_sharded_stream_equal_the_reference(workdir):
    """BASELINE config 4 (8 GiB over 8 GPUs) on one GPU: slice r is scanned exactly as rank r of `bench.py --gpus 8`
    scans it -- 1 GiB plus the maxPatternLen + 1 bytes of its successor (omp_PFAC.cpp:324), the last slice alone -- and
    its match count and position checksum equal the digest of the REFERENCE's PFAC_CPU_OMP result for that slice
    (tests/golden/full_digests.json: `inner` / `last`); folded in rank order they equal the folded reference digests,
    which is rank 0's check (omp_PFAC.cpp:396-439)."""
    n, world = 1 << 30, 8
    dg = digests("c3", 1024)
    assert sorted(dg) == list(range(world))
    cfg = wl.make_config("c3")
    pf = wl.write_pattern_file(f"{workdir}/c4.pat", cfg.patterns)
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_AUTOMATIC, api.PFACX_KERNEL_AUTO)
    max_len = h.info().maxPatternLen
    facts = []
    d_in = torch.empty(n + sharding.overlap_bytes(max_len), dtype=torch.uint8, device="cuda:0")
    d_out = torch.empty(n + sharding.overlap_bytes(max_len), dtype=torch.int32, device="cuda:0")
    try:
        with concurrent.futures.ThreadPoolExecutor(3) as pool:      # the generator is C code: it runs beside the scans
            jobs = [pool.submit(sharding.rank_input, cfg, n, r, world, max_len) for r in range(world)]
            for r in range(world):
                host, owned = jobs[r].result()
                jobs[r] = None
                assert owned == n and wl.fnv1a(host[:n]) == dg[r]["input_fnv1a"], "input generator drifted"
                d_in[: host.size].copy_(torch.from_numpy(host))
                d_out.fill_(-1)
                h.matchFromDevice(d_in.data_ptr(), host.size, d_out.data_ptr())
                torch.cuda.synchronize()
                pos = torch.nonzero(d_out[:n]).flatten()
                ids = d_out[:n][pos].cpu().numpy()
                pos = pos.cpu().numpy().astype(np.int64)
                assert int(d_out[:n].min()) >= 0, "a position was not written"
                want = dg[r]["last" if r == world - 1 else "inner"]
                got = (int(pos.size), sharding.position_checksum(pos, ids, base=r * n))
                assert got == (want["match_count"], want["checksum"]), (r, got, want)
                facts.append((got[0], got[1] & 0x7FFFFFFFFFFFFFFF))
                del host
    finally:
        h.destroy()
    folded = sharding.combine_checksums(facts)
    expected = sharding.combine_checksums([(dg[r]["last" if r == world - 1 else "inner"]["match_count"],
                                            dg[r]["last" if r == world - 1 else "inner"]["checksum"] & 0x7FFFFFFFFFFFFFFF) for r in range(world)])
    assert folded == expected


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


def test_input_larger_than_4_gib(workdir):
    """The vector kernel keeps positions in 32 bits; inputs of 4 GiB and more are scanned as
    consecutive windows (scan_module.hip: kMaxLaunchBytes) whose overlap is rewritten by the next
    window.  Patterns planted across the window boundary, across 2^32 and at the very end must be
    reported, and the whole vector must equal those of the tiled kernel (64-bit group offsets) and of the reference-shaped
    kernel (size_t positions, reference-layout table).
    ANCHOR: the reference's sizes are `int` (< 2 GiB), so there is no reference output at this size; this is three
    independent kernels against each other plus the planted patterns, whose expected IDs come from the pattern file.
    The same stream's first 1 GiB is pinned on the reference digest above (test_full_size_result_equals_reference_digest)."""
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

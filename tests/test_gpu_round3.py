"""GPU tests, third batch: the multi-GPU data path on one GPU -- every slice of the 8 GiB BASELINE stream (config 4)
against the digests of the REFERENCE's CPU/OMP output and folded the way rank 0 folds them; one RCCL all_gather on
hardware through the rank path of bench.py; the library's multi-GPU driver on every visible device -- and the two
cliffs round 2 left: pointers that are not 16-byte aligned, and input in which most positions match.

Reference models: PFAC/test/omp_PFAC.cpp:257-439 (sliced run == single run, one context per device).
"""

import concurrent.futures
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from pfac_amd import api, sharding  # noqa: E402
from pfac_amd import workloads as wl  # noqa: E402
from tests.test_gpu_parity import make_handle  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available(), "GPU tests need a device; there is no CPU fallback to test"
    torch.cuda.set_device(0)


def _digests(workload, size_mib):
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "full_digests.json")))
    return {r["slice"]: r for r in doc["records"] if r["workload"] == workload and r["size_mib"] == size_mib}


def test_all_eight_slices_of_the_sharded_stream_equal_the_reference(workdir):
    """BASELINE config 4 (8 GiB over 8 GPUs) on one GPU: slice r is scanned exactly as rank r of `bench.py --gpus 8`
    scans it -- 1 GiB plus the maxPatternLen + 1 bytes of its successor (omp_PFAC.cpp:324), the last slice alone -- and
    its match count and position checksum equal the digest of the REFERENCE's PFAC_CPU_OMP result for that slice
    (tests/golden/full_digests.json: `inner` / `last`); folded in rank order they equal the folded reference digests,
    which is rank 0's check (omp_PFAC.cpp:396-439)."""
    n, world = 1 << 30, 8
    dg = _digests("c3", 1024)
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


def test_rank_path_runs_one_rccl_all_gather_on_hardware():
    """`bench.py --gpus 8` is eight of these processes: init_process_group("nccl") bound to the device, the facts
    all-gather on a DEVICE tensor, barriers around the timed region, destroy_process_group.  No multi-GPU node is
    needed to execute that code once: a world of one rank, in a fresh child process (never a re-exec of a process
    that has touched the GPU)."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29500 + os.getpid() % 2000), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist-backend", "nccl", "--force-dist", "--size-mib", "64",
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--pmc", "off"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["config"]["dist_backend"] == "nccl" and out["config"]["ranks_seen"] == [0]
    assert out["config"]["bit_exact"] is True and out["value"] > 0


def test_multi_gpu_driver_on_every_visible_device(workdir):
    """PFACX_matchFromHostMultiGPU with one worker per visible device: on a multi-GPU node this is
    the first launch of the 150 KiB-LDS kernel on devices 1..N-1 of a process (the launch attribute is per-device state);
    with one GPU it still runs the driver with its per-device handle.  Result == oracle."""
    from oracle import binding as ob
    ndev = torch.cuda.device_count()
    pats = wl.snort_patterns(3000)
    pf = wl.write_pattern_file(os.path.join(workdir, "mgpu3.pat"), pats)
    n = (24 << 20) + 333
    data = wl.http_stream(n, wl.http_message_pool(pats, pool_size=256, embed_fraction=0.3)).copy()
    o = ob.Oracle(pf, dense=False, hashed=True)
    want = o.match(data, hashed=True, omp=True)
    o.close()
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_AUTOMATIC)
    try:
        for devices in (list(range(ndev)), list(range(ndev)) * 2, list(range(ndev))):
            got = np.full(n, -3, dtype=np.int32)
            h.matchFromHostMultiGPU(data.ctypes.data, n, got.ctypes.data, devices)
            assert np.array_equal(got, want), f"devices {devices}"
            h.trim()                                              # PFACX_trim: the staging buffers come back on the next call
    finally:
        h.destroy()
        torch.cuda.set_device(0)


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
    assert min(rates.values()) >= 0.8 * rates[(0, 0)], rates


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
    the full-result path is what test_gpu_parity.py / test_gpu_round2.py pin on the oracle and the reference digests, and the
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
        assert 0.0 < h.scanStats()["filterKernelMs"] < 5.0
        h.setKernelTiming(False)
        h.matchFromDevice(d_in.data_ptr(), n, d_res.data_ptr())
        assert "filterKernelMs" not in h.scanStats()
    finally:
        h.destroy()

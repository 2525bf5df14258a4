"""The sharded path on hardware: what runs on ONE GPU of the N > 1 orchestration (strong scaling at N = 1, two ranks sharing device 0 over
gloo, one RCCL all_gather in a world of one, the library's multi-GPU drivers with several workers on one device), and the tests a
multi-GPU box runs first (skipped on one GPU).  Reference: PFAC/test/omp_PFAC.cpp:257-439, SimpleMultiGPU_pthread.cpp:50-174."""

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


two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")


def test_strong_scaling_mode_on_one_gpu():
    """`bench.py --scaling strong` at N = 1: the eight 8 MiB slices of the c3 stream on one device, each equal to its committed
    reference digest, folded like rank 0 folds them.  (The 8 GiB form is other_configs.c4_8gib_one_gpu of the default bench
    line; this one runs in seconds.)"""
    out = run_bench("--gpus", "1", "--scaling", "strong", "--total-mib", "64", "--size-mib", "8", "--steps", "3", "--warmup", "1")
    assert out["scaling"] == "strong" and out["n_gpus"] == 1 and out["config"]["slices_per_rank"] == 8
    assert out["config"]["bit_exact"] is True and out["config"]["folded_reference"]["equal"] is True      # 8 MiB slices have committed digests
    assert out["config"]["folded_result"]["match_count"] == 4579 + 4590 + 4579 + 4571 + 4454 + 4655 + 4583 + 4493
    assert out["value"] > 0 and out["config"]["kernel_launched"] == "pfac_scan_tiled"                       # 8 MiB calls: the tiled kernel (AUTO)


def test_two_ranks_on_one_device_over_gloo_weak_and_strong():
    """The orchestration the driver's N > 1 runs go through -- bench.py starting its own rank processes, every rank scanning its
    slice(s), rank 0 folding the gathered (count, checksum) facts against the committed reference digests -- with both ranks on
    THIS device (--dist-backend gloo: the ranks share GPU 0), so that none of it is executed for the first time on a multi-GPU
    node.  Weak: slices 0 and 1 of the c3 stream (8 MiB each: committed digests); strong: eight 8 MiB slices dealt round-robin."""
    weak = run_bench("--gpus", "2", "--dist-backend", "gloo", "--size-mib", "8", "--steps", "3", "--warmup", "1")
    assert weak["n_gpus"] == 2 and weak["config"]["ranks_seen"] == [0, 1] and weak["config"]["dist_backend"] == "gloo"
    assert weak["config"]["bit_exact"] is True and weak["scaling"] == "weak" and weak["config"]["folded_result"]["match_count"] == 4579 + 4590
    strong = run_bench("--gpus", "2", "--dist-backend", "gloo", "--scaling", "strong", "--total-mib", "64", "--size-mib", "8", "--steps", "3", "--warmup", "1")
    assert strong["scaling"] == "strong" and strong["n_gpus"] == 2 and strong["config"]["slices_per_rank"] == 4
    assert strong["config"]["ranks_seen"] == [0, 1]
    assert strong["config"]["bit_exact"] is True and strong["config"]["folded_reference"]["equal"] is True
    assert strong["config"]["folded_result"]["match_count"] == 4579 + 4590 + 4579 + 4571 + 4454 + 4655 + 4583 + 4493


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


@two_gpus
def test_two_rank_rccl_bench_weak_and_strong():
    """`bench.py --gpus 2` over RCCL, one GPU per rank: the weak line the driver runs (slices 0 and 1, each against its reference
    digest, facts all-gathered on device tensors) and the strong one (eight slices dealt round-robin over two ranks)."""
    weak = run_bench("--gpus", "2", "--size-mib", "1024", "--steps", "5", "--warmup", "2")
    assert weak["n_gpus"] == 2 and weak["config"]["ranks_seen"] == [0, 1] and weak["config"]["dist_backend"] == "nccl"
    assert weak["config"]["bit_exact"] is True and weak["config"]["folded_reference"]["equal"] is True
    assert weak["config"]["folded_result"]["match_count"] == 583306 + 581991                                  # tests/golden/full_digests.json
    strong = run_bench("--gpus", "2", "--scaling", "strong", "--total-gib", "8", "--steps", "3", "--warmup", "1")
    assert strong["scaling"] == "strong" and strong["n_gpus"] == 2 and strong["config"]["slices_per_rank"] == 4
    assert strong["config"]["bit_exact"] is True and strong["config"]["folded_reference"]["equal"] is True
    assert strong["config"]["bytes_total"] == 8 << 30


def test_multi_gpu_driver_on_one_device(workdir):
    """PFACX_matchFromHostMultiGPU: the library shards a host stream over the listed devices,
    one worker thread and one internal handle each (SimpleMultiGPU_pthread.cpp:50-174).  Listing device 0 several
    times exercises the whole path on a one-GPU box; the result must equal the oracle, including matches across
    the slice boundaries, for 1, 2 and 3 workers and on repeated calls (cached per-device handles)."""
    from oracle import binding as ob
    pats = wl.snort_patterns(3000)
    pf = wl.write_pattern_file(os.path.join(workdir, "multigpu.pat"), pats)
    n = (48 << 20) + 4321
    data = wl.http_stream(n, wl.http_message_pool(pats, pool_size=256, embed_fraction=0.3)).copy()
    longest = np.frombuffer(max(pats, key=len), dtype=np.uint8)
    for workers in (2, 3):
        for i in range(1, workers):
            cut = (n * i // workers) // 1024 * 1024
            data[cut - longest.size // 2: cut - longest.size // 2 + longest.size] = longest
    o = ob.Oracle(pf, dense=False, hashed=True)
    want = o.match(data, hashed=True, omp=True)
    o.close()
    h = make_handle(pf, api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_ON)
    try:
        for devices in ([0], [0, 0], [0, 0, 0], None, [0, 0]):
            got = np.full(n, -7, dtype=np.int32)
            h.matchFromHostMultiGPU(data.ctypes.data, n, got.ctypes.data, devices)
            assert_same(got, want, f"multi-GPU driver, devices {devices}")
        if torch.cuda.device_count() <= 7:
            assert h.matchFromHostMultiGPU(data.ctypes.data, n, got.ctypes.data, [7], check=False) == api.STATUS.INVALID_PARAMETER
        assert_same(device_match(h, data[: 1 << 20]), o_prefix(pf, data[: 1 << 20]), "the handle itself still works")
    finally:
        h.destroy()


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


@two_gpus
def test_multi_gpu_driver_on_devices_0_and_1(tmp_path):
    """PFACX_matchFromHostMultiGPU on devices [0, 1]: two worker threads, two internal handles, the first launches of the
    160 KiB-LDS kernels on device 1 of this process; matches across the slice boundary; a later change of the parent's modes
    reaches the cached children.  Result == oracle."""
    from oracle import binding as ob
    pats = wl.snort_patterns(3000)
    pf = wl.write_pattern_file(str(tmp_path / "mgpu2.pat"), pats)
    n = (80 << 20) + 777                                             # two pieces per device, the filter kernel on both
    data = wl.http_stream(n, wl.http_message_pool(pats, pool_size=256, embed_fraction=0.3)).copy()
    cut = (n // 2) // 1024 * 1024
    p = np.frombuffer(pats[5], dtype=np.uint8)
    data[cut - 3:cut - 3 + p.size] = p                               # a pattern across the boundary of the two slices
    o = ob.Oracle(pf, dense=False, hashed=True)
    want = o.match(data, hashed=True, omp=True)
    o.close()
    h = api.PFAC.create()
    h.setPerfMode(api.PFAC_SPACE_DRIVEN)
    h.readPatternFromFile(pf)
    try:
        for devices, variant in (([0, 1], api.PFACX_KERNEL_AUTO), ([1, 0], api.PFACX_KERNEL_NAIVE), ([0, 1, 1], api.PFACX_KERNEL_FILTER)):
            h.setKernelVariant(variant)                               # the children of the previous call pick it up
            got = np.full(n, -3, dtype=np.int32)
            h.matchFromHostMultiGPU(data.ctypes.data, n, got.ctypes.data, devices)
            assert np.array_equal(got, want), f"devices {devices}"
        # the compacted-output form over the same devices: the pairs of the whole stream in position order
        want_pos = np.flatnonzero(want)
        ids, pos = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        for devices in ([0, 1], [1, 0, 1]):
            _, count = h.matchFromHostReduceMultiGPU(data.ctypes.data, n, ids.ctypes.data, pos.ctypes.data, devices)
            assert count == want_pos.size and np.array_equal(pos[:count], want_pos) and np.array_equal(ids[:count], want[want_pos]), f"reduce, devices {devices}"
    finally:
        h.destroy()
        torch.cuda.set_device(0)


def test_match_from_host_reduce_over_several_workers(workdir):
    """PFACX_matchFromHostReduceMultiGPU: the compacted-output call sharded over worker threads / per-device handles (here every worker on
    device 0: one, two and three slices; test_multi_gpu_driver_on_devices_0_and_1 runs devices [0, 1] where there are two).  Matches across every slice
    boundary and at the very end of the stream; slices of several pieces; the pairs of the whole stream in position order == the non-zero
    entries of the oracle's result (reference model: PFAC/test/omp_PFAC.cpp:257-439 + PFAC.cpp:1010-1128)."""
    pats = wl.snort_patterns(3000)
    pf = wl.write_pattern_file(os.path.join(workdir, "mgpu_reduce.pat"), pats)
    n = (70 << 20) + 333                                              # more than one 16 Mi-position piece per worker
    data = wl.http_stream(n, wl.http_message_pool(pats, pool_size=256, embed_fraction=0.3)).copy()
    p = np.frombuffer(pats[7], dtype=np.uint8)
    for cut in ((n // 2) // 1024 * 1024, (n // 3) // 1024 * 1024, (2 * n // 3) // 1024 * 1024):
        data[cut - 2:cut - 2 + p.size] = p                            # a pattern across the boundary of two slices
    data[n - p.size:] = p                                             # ... and one that ends with the stream
    want = oracle_match(pf, data, omp=True)
    want_pos = np.flatnonzero(want)
    h = api.PFAC.create()
    h.setPerfMode(api.PFAC_SPACE_DRIVEN)
    h.readPatternFromFile(pf)
    try:
        ids, pos = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        for devices in ([0], [0, 0], [0, 0, 0], None):
            ids.fill(-7)
            pos.fill(-7)
            _, count = h.matchFromHostReduceMultiGPU(data.ctypes.data, n, ids.ctypes.data, pos.ctypes.data, devices)
            assert count == want_pos.size, (devices, count, want_pos.size)
            assert np.array_equal(pos[:count], want_pos) and np.array_equal(ids[:count], want[want_pos]), devices
        _, single = h.matchFromHostReduce(data.ctypes.data, n, ids.ctypes.data, pos.ctypes.data)
        assert single == want_pos.size and np.array_equal(pos[:single], want_pos)
    finally:
        h.destroy()


def test_cpp_multi_gpu_example(workdir):
    """examples/multi_gpu.cpp: PFACX_matchFromHostMultiGPU from C++ with three workers, self-checked against a
    single-device scan the way PFAC/test/omp_PFAC.cpp:396-439 checks its sliced run."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-B"], stdout=subprocess.DEVNULL)
    pats = wl.snort_patterns(1500)
    pf = wl.write_pattern_file(os.path.join(workdir, "cpp_multi.pat"), pats)
    data = wl.http_stream((9 << 20) + 77, wl.http_message_pool(pats, pool_size=128, embed_fraction=0.3))
    inp = os.path.join(workdir, "cpp_multi.in")
    data.tofile(inp)
    out = subprocess.run([os.path.join(ROOT, "examples", "multi_gpu"), pf, inp, "3"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 differences" in out.stdout and "3 worker(s)" in out.stdout
    assert "equal to the non-zero entries of the full result" in out.stdout, out.stdout          # PFACX_matchFromHostReduceMultiGPU

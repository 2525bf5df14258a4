"""GPU tests that need MORE THAN ONE device (skipped on a one-GPU box, so that the first multi-GPU box that runs the suite
executes every multi-GPU line of the repo): the 2-rank RCCL bench in both scaling modes, the library's multi-GPU driver
on devices [0, 1]; and what runs everywhere: the strong-scaling mode on one device, two ranks SHARING one device over gloo
(the orchestration of the N > 1 runs, minus RCCL).

Reference model: PFAC/test/omp_PFAC.cpp:257-439 (one context per device, slices dealt round-robin with a
max_patternLen + 1 overlap, the folded result of all slices equals the single run's)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from pfac_amd import api  # noqa: E402
from pfac_amd import workloads as wl  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")


def _bench(*flags, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags, "--no-cpu-baseline", "--no-other-configs", "--pmc", "off"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    return json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])


def test_strong_scaling_mode_on_one_gpu():
    """`bench.py --scaling strong` at N = 1: the eight 8 MiB slices of the c3 stream on one device, each equal to its committed
    reference digest, folded like rank 0 folds them.  (The 8 GiB form is other_configs.c4_8gib_one_gpu of the default bench
    line; this one runs in seconds.)"""
    out = _bench("--gpus", "1", "--scaling", "strong", "--total-mib", "64", "--size-mib", "8", "--steps", "3", "--warmup", "1")
    assert out["scaling"] == "strong" and out["n_gpus"] == 1 and out["config"]["slices_per_rank"] == 8
    assert out["config"]["bit_exact"] is True and out["config"]["folded_reference"]["equal"] is True      # 8 MiB slices have committed digests
    assert out["config"]["folded_result"]["match_count"] == 4579 + 4590 + 4579 + 4571 + 4454 + 4655 + 4583 + 4493
    assert out["value"] > 0 and out["config"]["kernel_launched"] == "pfac_scan_tiled"                       # 8 MiB calls: the tiled kernel (AUTO)


def test_two_ranks_on_one_device_over_gloo_weak_and_strong():
    """The orchestration the driver's N > 1 runs go through -- bench.py starting its own rank processes, every rank scanning its
    slice(s), rank 0 folding the gathered (count, checksum) facts against the committed reference digests -- with both ranks on
    THIS device (--dist-backend gloo: the ranks share GPU 0), so that none of it is executed for the first time on a multi-GPU
    node.  Weak: slices 0 and 1 of the c3 stream (8 MiB each: committed digests); strong: eight 8 MiB slices dealt round-robin."""
    weak = _bench("--gpus", "2", "--dist-backend", "gloo", "--size-mib", "8", "--steps", "3", "--warmup", "1")
    assert weak["n_gpus"] == 2 and weak["config"]["ranks_seen"] == [0, 1] and weak["config"]["dist_backend"] == "gloo"
    assert weak["config"]["bit_exact"] is True and weak["scaling"] == "weak" and weak["config"]["folded_result"]["match_count"] == 4579 + 4590
    strong = _bench("--gpus", "2", "--dist-backend", "gloo", "--scaling", "strong", "--total-mib", "64", "--size-mib", "8", "--steps", "3", "--warmup", "1")
    assert strong["scaling"] == "strong" and strong["n_gpus"] == 2 and strong["config"]["slices_per_rank"] == 4
    assert strong["config"]["ranks_seen"] == [0, 1]
    assert strong["config"]["bit_exact"] is True and strong["config"]["folded_reference"]["equal"] is True
    assert strong["config"]["folded_result"]["match_count"] == 4579 + 4590 + 4579 + 4571 + 4454 + 4655 + 4583 + 4493


@two_gpus
def test_two_rank_rccl_bench_weak_and_strong():
    """`bench.py --gpus 2` over RCCL, one GPU per rank: the weak line the driver runs (slices 0 and 1, each against its reference
    digest, facts all-gathered on device tensors) and the strong one (eight slices dealt round-robin over two ranks)."""
    weak = _bench("--gpus", "2", "--size-mib", "1024", "--steps", "5", "--warmup", "2")
    assert weak["n_gpus"] == 2 and weak["config"]["ranks_seen"] == [0, 1] and weak["config"]["dist_backend"] == "nccl"
    assert weak["config"]["bit_exact"] is True and weak["config"]["folded_reference"]["equal"] is True
    assert weak["config"]["folded_result"]["match_count"] == 583306 + 581991                                  # tests/golden/full_digests.json
    strong = _bench("--gpus", "2", "--scaling", "strong", "--total-gib", "8", "--steps", "3", "--warmup", "1")
    assert strong["scaling"] == "strong" and strong["n_gpus"] == 2 and strong["config"]["slices_per_rank"] == 4
    assert strong["config"]["bit_exact"] is True and strong["config"]["folded_reference"]["equal"] is True
    assert strong["config"]["bytes_total"] == 8 << 30


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

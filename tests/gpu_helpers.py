"""Helpers shared by the GPU test files (tests/test_full_result.py, test_reduce.py, test_host_paths.py, test_hostile.py,
test_kernel_variants.py, test_full_size_digests.py, test_multi_gpu.py): handles in every table mode and kernel variant, poisoned
device buffers, the committed reference digests, event-timed launches.  Test infrastructure only."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from pfac_amd import api  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODES = [
    (api.PFAC_TIME_DRIVEN, api.PFAC_TEXTURE_OFF, "dense-global"),
    (api.PFAC_TIME_DRIVEN, api.PFAC_TEXTURE_ON, "dense-buffer"),
    (api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_OFF, "hash-global"),
    (api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_ON, "hash-buffer"),
]
STAGE = api.PFACX_WALKER_STAGE << 8             # make_handle: the walker of the full-result filter kernel rides in the variant's second byte
VETO = api.PFACX_WALKER_VETO << 8
VARIANTS = [(api.PFACX_KERNEL_FILTER, "filter"), (api.PFACX_KERNEL_FILTER | STAGE, "filter-stage"), (api.PFACX_KERNEL_FILTER | VETO, "filter-veto"),
            (api.PFACX_KERNEL_NAIVE, "naive"), (api.PFACX_KERNEL_AUTO, "auto"), (api.PFACX_KERNEL_REFTABLE, "reftable")]
WALKERS = [(api.PFACX_WALKER_WINDOW, "window"), (api.PFACX_WALKER_STAGE, "stage")]


def perf_asserts():
    """Assertions that compare wall-clock or event times (one kernel variant against another, pinned against pageable buffers, a GB/s
    floor) run only when PFAC_PERF_FLOORS is set: the default `-m gpu` run asserts result equality and nothing else -- like the
    reference's own self-check (PFAC/test/omp_PFAC.cpp:420-439) -- so that a busy or power-capped box cannot turn it red.  The rates
    are printed either way."""
    return bool(os.environ.get("PFAC_PERF_FLOORS"))


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


def digest_record(workload, slice_index, size_mib):
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "full_digests.json")))
    for r in doc["records"]:
        if (r["workload"], r["slice"], r["size_mib"]) == (workload, slice_index, size_mib):
            return r
    raise KeyError((workload, slice_index, size_mib))


def digests(workload, size_mib):
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "full_digests.json")))
    return {r["slice"]: r for r in doc["records"] if r["workload"] == workload and r["size_mib"] == size_mib}


def timed_match(h, data, steps=5):
    from pfac_amd import hiprt
    n = int(data.size)
    d_in = torch.from_numpy(np.ascontiguousarray(data)).to("cuda:0")
    d_out = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
    h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
    torch.cuda.synchronize()
    a, b = hiprt.Event(), hiprt.Event()
    a.record(0)
    for _ in range(steps):
        h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
    b.record(0)
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), n / (a.elapsed_ms(b) / steps / 1e3) / 1e9


def oracle_match(pf, data, omp=False):
    from oracle import binding as ob
    o = ob.Oracle(pf, hashed=False)
    want = o.match(data, omp=omp)
    o.close()
    return want


def run_bench(*flags, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags, "--no-cpu-baseline", "--no-other-configs", "--pmc", "off"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    return json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])

def o_prefix(pf, data):
    from oracle import binding as ob
    o = ob.Oracle(pf, dense=False, hashed=True)
    try:
        return o.match(data, hashed=True, omp=True)
    finally:
        o.close()

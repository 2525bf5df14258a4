"""Shared fixtures.

``-m "not gpu"`` : oracle vs golden vectors, host logic of the C ABI (through a
host-only handle), symbol checks, gloo sharding tests.  No device needed.
``-m gpu``       : parity tests proper -- every call goes through the C ABI into
the HIP kernels on cuda:0 and is compared bit-for-bit with the oracle.
"""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _device(request):
    """Every test marked `gpu` runs on cuda:0 and fails -- not skips -- without one: there is no CPU fallback to test."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        assert torch.cuda.is_available(), "GPU tests need a device; there is no CPU fallback to test"
        torch.cuda.set_device(0)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the product libraries, the workload generator and the oracle if they are missing."""
    import __graft_entry__ as entry
    entry.build(only_if_missing=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("pfac"))


class Workload:
    """A pattern file on disk plus an input stream (numpy uint8)."""

    def __init__(self, name, pattern_file, data):
        self.name = name
        self.pattern_file = pattern_file
        self.data = data


@pytest.fixture(scope="session")
def workloads(workdir):
    """Small instances of the BASELINE.json configurations (oracle finishes in seconds)."""
    from pfac_amd import workloads as wl

    out = {}

    def add(name, pats, data):
        pf = wl.write_pattern_file(os.path.join(workdir, name + ".pat"), pats)
        out[name] = Workload(name, pf, np.ascontiguousarray(data, dtype=np.uint8))

    add("c1", wl.example_patterns(), np.frombuffer(wl.example_input(), dtype=np.uint8))
    add("ex2", wl.example2_patterns(), np.frombuffer(wl.example2_input(), dtype=np.uint8))
    # C2 small: random patterns over random bytes, with every pattern planted once so matches exist
    p2 = wl.random_patterns()
    d2 = wl.random_bytes((1 << 20) + 37).copy()
    rng = np.random.Generator(np.random.PCG64(11))
    for p in p2[:200]:
        at = int(rng.integers(0, d2.size - 64))
        d2[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
    add("c2", p2, d2)
    p3 = wl.snort_patterns(3000)
    add("c3", p3, wl.http_stream((1 << 20) + 1, wl.http_message_pool(p3, pool_size=512, embed_fraction=0.2)))
    p5 = wl.adversarial_patterns(200)
    add("c5", p5, wl.adversarial_stream(1 << 18, wl.adversarial_pool(p5, pool_size=256)))
    # dense matches: every position of a run matches something (stresses the patch-store path)
    pa = [b"a", b"aa", b"aaa", b"aaaa", b"ab", b"b" * 7, b"abc" * 5]
    da = np.frombuffer((b"a" * 3000 + b"b" * 50 + b"abc" * 400 + b"aab") * 3, dtype=np.uint8)
    add("dense_hits", pa, da)
    # bytes >= 0x80 and NUL: signed-char ordering of the pattern sort, binary-safe matching
    pb = [bytes([0xFF, 0x80, 0x00]), bytes([0x00, 0x00]), bytes([0x7F, 0x80]), bytes([0x80]), b"\x01\x02\x03\x04\x05",
          bytes([0xFF, 0x80, 0x00, 0x41])]
    db = np.random.Generator(np.random.PCG64(5)).choice(
        np.array([0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x41, 0x7F, 0x80, 0xFF], dtype=np.uint8), size=70001)
    add("binary", pb, db)
    return out


@pytest.fixture(scope="session")
def oracle_results(workloads):
    """Oracle (dense, scalar) result for every small workload, computed once."""
    from oracle import binding as ob
    res = {}
    for name, w in workloads.items():
        o = ob.Oracle(w.pattern_file, hashed=False)
        res[name] = o.match(w.data)
        o.close()
    return res

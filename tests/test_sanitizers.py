"""The host library under sanitizers (CPU only; GPU AddressSanitizer is not available on the pool).

`make -C pfac_amd/csrc san tsan` builds libpfac.so with AddressSanitizer + UBSan and with ThreadSanitizer.  The host code is
what parses untrusted bytes (PFACX_loadCompiled, the four pattern readers) and what several host threads share (the handle's
locks), so:
  * tests/test_host_api.py runs against the ASan + UBSan build (the library is loaded into an uninstrumented python: the
    runtime is preloaded);
  * tools/fuzz_host.cpp mutates compiled sets (checksum recomputed) and pattern text in a loop and matches with whatever is
    accepted -- a few thousand iterations here, 10^5 in `make -C pfac_amd/csrc fuzz` (profiles/r05_fuzz_host.txt);
  * tools/tsan_host.cpp shares one handle between threads that match on the CPU platform while one of them changes the perf
    mode (round 5: that found PFAC_matchFromHost on the CPU platforms reading tables PFAC_setPerfMode was rebuilding).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pfac_amd", "csrc")


@pytest.fixture(scope="module")
def sanitizer_builds():
    p = subprocess.run(["make", "-C", CSRC, "san", "tsan"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    return os.path.join(ROOT, "pfac_amd", "lib", "san", "libpfac.so"), os.path.join(CSRC, "build", "fuzz_host"), os.path.join(CSRC, "build", "tsan_host")


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(path):
        pytest.skip(name + " not found next to gcc")
    return path


def test_host_api_tests_under_asan_and_ubsan(sanitizer_builds):
    lib, _, _ = sanitizer_builds
    env = dict(os.environ, PFAC_HOST_LIB=lib, LD_PRELOAD=_runtime("libasan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_api.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = p.stdout[-4000:]
    assert p.returncode == 0 and "AddressSanitizer" not in p.stdout and "runtime error" not in p.stdout, tail
    assert " passed" in tail


def test_mutated_compiled_sets_and_pattern_files(sanitizer_builds, tmp_path):
    _, fuzz, _ = sanitizer_builds
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    for seed in (1, 20261002):
        p = subprocess.run([fuzz, "4000", str(seed), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert p.returncode == 0 and "no sanitizer report" in p.stdout and "AddressSanitizer" not in p.stdout and "runtime error" not in p.stdout, p.stdout[-3000:]


def test_shared_handle_under_tsan(sanitizer_builds):
    _, _, tsan = sanitizer_builds
    p = subprocess.run([tsan, "6", "40", os.path.join(ROOT, "tests", "golden", "example_pattern")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "ThreadSanitizer" not in p.stdout and "0 wrong results" in p.stdout, p.stdout[-3000:]

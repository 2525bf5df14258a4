"""The filter kernel keeps the chunk in flight in vector registers that the compiler is told not to use
(scan_filter.hip: prefetchChunk / kCompilerVgprs).  The contract is checked by the BUILD (pfac_amd/csrc/Makefile runs
pfac_amd/csrc/check_isa.py on the ISA it has just generated and fails without a library); these tests run the same
functions on a fresh compile, and show that a kernel compiled for too many registers is refused.  hipcc on the CPU: no
GPU needed."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
_spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "pfac_amd", "csrc", "check_isa.py"))
check_isa = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(check_isa)


def _compile(out, *flags):
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", *flags,
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pfac_amd", "csrc"),
           os.path.join(ROOT, "pfac_amd", "csrc", "scan_filter.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text()


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    return _compile(tmp_path_factory.mktemp("isa") / "scan.s")


def test_filter_kernel_register_contract(isa):
    check_isa.check_registers(isa)


def test_scan_loop_waits_once_per_trip(isa):
    check_isa.check_waits(isa)


def test_a_broken_register_budget_is_refused(tmp_path):
    """-DPFAC_COMPILER_VGPRS=48: a budget the kernel does not fit (it spills: a scratch reload is a vector-memory load the
    hand-written wait does not count) -- what a toolchain bump that allocates differently would look like.  check_isa.py, and
    with it `make` (which then leaves no libpfac_gfx950.so behind), must say no."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    text = _compile(tmp_path / "broken.s", "-DPFAC_COMPILER_VGPRS=48", "-DPFAC_QUICK")
    saved = check_isa.INSTANCES
    check_isa.INSTANCES = 2                                           # the quick build has the two bench instances only
    try:
        with pytest.raises(check_isa.ContractError):
            check_isa.check_registers(text)
    finally:
        check_isa.INSTANCES = saved

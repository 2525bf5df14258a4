"""The filter kernel keeps the chunk in flight in vector registers that the compiler is told not to use
(scan_gfx950.hip: prefetchChunk / kCompilerVgprs).  That contract is between the source and one compiler version, so
it is checked on the generated ISA: the reserved registers appear only as destinations of the prefetch loads and as
sources of the copies that take a tile, every instance still owns 128 registers, and nothing spills (a scratch reload
is a vector-memory load the hand-written wait does not count).  Runs hipcc on the CPU: no GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
RESERVED = {f"v{i}" for i in range(119, 128)}


def _registers(operand):
    """v5 -> {v5}; v[120:123] -> {v120..v123}"""
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", operand):
        if m.group(3) is not None:
            out.add(f"v{m.group(3)}")
        else:
            out.update(f"v{i}" for i in range(int(m.group(1)), int(m.group(2)) + 1))
    return out


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("isa") / "scan.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pfac_amd", "csrc"),
           os.path.join(ROOT, "pfac_amd", "csrc", "scan_gfx950.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text()


def _kernels(text):
    """name -> (body, descriptor) of every pfac_scan_filter instance"""
    found = {}
    for m in re.finditer(r"^(_ZN\S*pfac_scan_filter\S*):.*?^\.Lfunc_end\d+:", text, re.S | re.M):
        name = m.group(1)
        d = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
        found[name] = (m.group(0), d.group(1) if d else "")
    return found


def test_filter_kernel_register_contract(isa):
    kernels = _kernels(isa)
    assert len(kernels) == 8, sorted(kernels)          # TEX x HAS_SHORT x REDUCE
    for name, (body, desc) in kernels.items():
        assert re.search(r"\.amdhsa_next_free_vgpr 128\b", desc), name
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", desc), (name, "the kernel spills")
        assert "scratch_" not in body, name
        loads = copies = 0
        for line in body.splitlines():
            code = line.split(";")[0].strip()
            if not code or code.startswith(".") or code.endswith(":"):
                continue
            used = _registers(code) & RESERVED
            if not used:
                continue
            op, _, rest = code.partition(" ")
            operands = [o.strip() for o in rest.split(",")]
            if op in ("global_load_dwordx4", "global_load_dword"):
                assert _registers(operands[0]) <= RESERVED and not (_registers(",".join(operands[1:])) & RESERVED), (name, code)
                loads += 1
            elif op == "v_mov_b32":
                assert not (_registers(operands[0]) & RESERVED) and _registers(operands[1]) <= RESERVED, (name, code)
                copies += 1
            else:
                raise AssertionError(f"{name}: reserved register in `{code}`")
        assert loads == 6 and copies == 10, (name, loads, copies)      # two prefetch sites x 3 loads; 5 + 5 copies


def test_scan_loop_waits_once_per_trip(isa):
    """Inside the scan loop the only wait for vector memory is the explicit one at the top of a trip (and, in the
    compacted-output instances, the ones behind the returning atomics that hand out chunks and output slots)."""
    for name, (body, _) in _kernels(isa).items():
        lines = body.splitlines()
        # the scan loop: from the first copy out of a reserved register back to the enclosing loop header
        first_copy = next(i for i, l in enumerate(lines) if re.search(r"v_mov_b32 v\d+, v120\b", l))
        header = max(i for i, l in enumerate(lines[:first_copy]) if "Loop Header: Depth=1" in l)
        label = re.match(r"\.L(BB\d+_\d+):", lines[header]).group(1)
        member = [i for i, l in enumerate(lines) if f"Header={label} " in l]             # blocks annotated as part of the loop
        end = next(i for i, l in enumerate(lines) if i > max(member) and re.match(r"\.LBB\d+_\d+:", l))
        # the compiler may place blocks of the loop (its rotated top, with the explicit wait) in front of the header label
        start = min(header, max(i for i, l in enumerate(lines[:min(member)]) if re.match(r"(\.LBB\d+_\d+:|; %bb\.\d+:)", l)))
        loop = lines[start:end]
        waits = [i for i, l in enumerate(loop) if "s_waitcnt" in l and "vmcnt" in l]
        # the compacted-output instances flush their staged pairs with a returning atomic now and then and wait for it
        after_atomic = [i for i in waits if any("global_atomic_add" in l for l in loop[max(0, i - 4):i])]
        # ... of which the compiler may lay out one copy per path into the loop top (tail duplication): every copy is
        # followed by the same instruction, the first of the walkers' consume stage
        top = {loop[i + 1].strip() for i in waits if i not in after_atomic}
        assert len(top) == 1 and 1 <= len(waits) - len(after_atomic) <= 2, (name, [loop[i].strip() + " / " + loop[i + 1].strip() for i in waits])
        if re.search(r"ELb0ELi\dE", name):                        # REDUCE = false: one atomic, the append to the list of pattern-dense chunks
            assert len(after_atomic) <= 1, name

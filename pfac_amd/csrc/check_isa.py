#!/usr/bin/env python3
"""check_isa.py <device ISA of scan_*.hip>  -- the register contract of the filter kernel, checked on what the compiler made.

The filter kernel keeps the chunk in flight in nine vector registers that the compiler is told not to use
(scan_*.hip: prefetchChunk / kCompilerVgprs: v119..v127, inline assembly).  That contract is between the source and
ONE compiler version, so the build checks it on the generated ISA and fails if it does not hold (pfac_amd/csrc/Makefile;
tests/test_kernel_isa.py runs the same functions):
  * every pfac_scan_filter instance owns 128 vector registers and has no scratch (a scratch reload is a vector-memory
    load the hand-written wait does not count);
  * the reserved registers appear only as destinations of the prefetch loads (three per site) and as sources of the ten
    copies that take a tile;
  * inside the scan loop the only wait for vector memory is the explicit one at the top of a trip (and, in the
    compacted-output instances, the ones behind the returning atomics that hand out chunks and output slots, and the
    ones behind the on-demand fetch of a long slot's extension unit / of input beyond a walk's LDS stage).
Exit status 0 = the contract holds; otherwise the first violation is printed."""
import re
import sys

RESERVED = {f"v{i}" for i in range(119, 128)}
INSTANCES = 20                                          # TEX x HAS_SHORT x {compacted output, full result with the window walker, ... with the stage walker, ... window walker + veto (tail table in LDS), ... (in device memory)}


class ContractError(Exception):
    pass


def _registers(operand):
    """v5 -> {v5}; v[120:123] -> {v120..v123}"""
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", operand):
        if m.group(3) is not None:
            out.add(f"v{m.group(3)}")
        else:
            out.update(f"v{i}" for i in range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def kernels(text):
    """name -> (body, descriptor) of every pfac_scan_filter instance"""
    found = {}
    for m in re.finditer(r"^(_ZN\S*pfac_scan_filter\S*):.*?^\.Lfunc_end\d+:", text, re.S | re.M):
        name = m.group(1)
        d = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
        found[name] = (m.group(0), d.group(1) if d else "")
    return found


def check_registers(text):
    ks = kernels(text)
    if len(ks) != INSTANCES:
        raise ContractError(f"{len(ks)} pfac_scan_filter instances, expected {INSTANCES}: {sorted(ks)}")
    for name, (body, desc) in ks.items():
        if not re.search(r"\.amdhsa_next_free_vgpr 128\b", desc):
            raise ContractError(f"{name}: does not own 128 vector registers")
        if not re.search(r"\.amdhsa_private_segment_fixed_size 0\b", desc) or "scratch_" in body:
            raise ContractError(f"{name}: the kernel spills")
        loads = copies = 0
        for line in body.splitlines():
            code = line.split(";")[0].strip()
            if not code or code.startswith(".") or code.endswith(":"):
                continue
            if not (_registers(code) & RESERVED):
                continue
            op, _, rest = code.partition(" ")
            operands = [o.strip() for o in rest.split(",")]
            if op in ("global_load_dwordx4", "global_load_dword"):
                if not (_registers(operands[0]) <= RESERVED) or (_registers(",".join(operands[1:])) & RESERVED):
                    raise ContractError(f"{name}: reserved register misused in `{code}`")
                loads += 1
            elif op == "v_mov_b32":
                if (_registers(operands[0]) & RESERVED) or not (_registers(operands[1]) <= RESERVED):
                    raise ContractError(f"{name}: reserved register misused in `{code}`")
                copies += 1
            else:
                raise ContractError(f"{name}: reserved register in `{code}` (the compiler allocated it: kCompilerVgprs is too large)")
        # prefetch sites x 3 loads: in front of the loop and behind level 1; the full-result instances fetch the chunk in flight once
        # more when a wave changes to stage mode (wider halo); 5 + 5 copies
        want_loads = 9 if re.search(r"ELb0ELi\dELb1E", name) else 6        # REDUCE = false, STAGE = true
        if (loads, copies) != (want_loads, 10):
            raise ContractError(f"{name}: {loads} prefetch loads and {copies} copies of reserved registers, expected {want_loads} and 10")


def check_waits(text):
    for name, (body, _) in kernels(text).items():
        lines = body.splitlines()
        # the scan loop: from the first copy out of a reserved register back to the enclosing loop header
        first_copy = next(i for i, l in enumerate(lines) if re.search(r"v_mov_b32 v\d+, v120\b", l))
        header = max(i for i, l in enumerate(lines[:first_copy]) if "Loop Header: Depth=1" in l)
        while not re.match(r"\.L(BB\d+_\d+):", lines[header]):      # the annotation may sit on the line(s) behind the label
            header -= 1
        label = re.match(r"\.L(BB\d+_\d+):", lines[header]).group(1)
        member = [i for i, l in enumerate(lines) if f"Header={label} " in l]             # blocks annotated as part of the loop
        end = next(i for i, l in enumerate(lines) if i > max(member) and re.match(r"\.LBB\d+_\d+:", l))
        # the compiler may place blocks of the loop (its rotated top, with the explicit wait) in front of the header label
        start = min(header, max(i for i, l in enumerate(lines[:min(member) + 1]) if re.match(r"(\.LBB\d+_\d+:|; %bb\.\d+:)", l)))     # + 1: the annotation may sit on the label's own line
        loop = lines[start:end]
        waits = [i for i, l in enumerate(loop) if "s_waitcnt" in l and "vmcnt" in l]
        # the compacted-output instances flush their staged pairs with a returning atomic now and then and wait for it
        after_atomic = [i for i in waits if any("global_atomic_add" in l for l in loop[max(0, i - 4):i])]
        # the compacted-output instances fetch the extension unit of a LONG slot (wide buckets, pfac_context.h) and the input
        # behind it when a header's first eight chain bytes have matched, and wait for them on the spot (`; pfac_ext_sync`
        # in the source); the full-result instances load the input of a walk that has run off its LDS stage -- patterns
        # longer than ~100 bytes -- the same way (`; pfac_deep_sync`).  Both are rare paths behind a wave-wide test.  (The VETO = 2
        # instances wait for the buckets of the device-memory tail table nowhere but at the top of the next trip.)
        def behind_marker(i):
            for j in range(i - 1, max(0, i - 60), -1):
                if "pfac_ext_sync" in loop[j] or "pfac_deep_sync" in loop[j]:
                    return True
            return False
        ext_sync = [i for i in waits if i not in after_atomic and behind_marker(i)]
        if len(ext_sync) > 4:                           # two marked places per instance (full-result: a step's bytes, a long slot's; compacted output: two walk sets), two loads each
            raise ContractError(f"{name}: {len(ext_sync)} on-the-spot waits behind pfac_ext_sync / pfac_deep_sync markers")
        waits = [i for i in waits if i not in ext_sync]
        # ... of which the compiler may lay out one copy per path into the loop top (tail duplication): every copy is
        # followed by the same instruction, the first of the walkers' consume stage
        top = {loop[i + 1].strip() for i in waits if i not in after_atomic}
        if not (len(top) == 1 and 1 <= len(waits) - len(after_atomic) <= 2):
            raise ContractError(f"{name}: waits for vector memory inside the scan loop: " + "; ".join(loop[i].strip() + " / " + loop[i + 1].strip() for i in waits))
        if re.search(r"ELb0ELi\dELb[01]E", name) and len(after_atomic) > 1:      # REDUCE = false: one atomic, the append to the list of pattern-dense chunks
            raise ContractError(f"{name}: {len(after_atomic)} waits behind atomics in the full-result scan loop")


def main(argv):
    if len(argv) != 2:
        print(__doc__)
        return 2
    text = open(argv[1]).read()
    try:
        check_registers(text)
        check_waits(text)
    except ContractError as e:
        print("check_isa.py: the filter kernel's register contract does not hold with this compiler / source:\n  " + str(e), file=sys.stderr)
        return 1
    print(f"check_isa.py: register contract holds for {INSTANCES} pfac_scan_filter instances")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))

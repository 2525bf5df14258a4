"""tools/hostile_driver.py <allmatch|snortlen> <variant: naive|filter|auto|reftable> [launches] [MiB]   (GPU box only)
A few PFAC_matchFromDevice launches over one of the two pattern-dense test inputs (tests/test_hostile.py), for profiling:
put `python3 tools/hostile_driver.py ...` directly behind `rocprofv3 ... --`."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfac_amd import api, workloads as wl

case, variant = sys.argv[1], sys.argv[2]
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 4
mib = int(sys.argv[4]) if len(sys.argv) > 4 else 64
n = mib << 20
if case == "allmatch":
    pats = [b"a" * k for k in range(1, 9)]
    data = np.full(n, ord("a"), dtype=np.uint8)
else:                                                         # the set and the text of test_snort_length_distribution_with_1_and_2_byte_patterns
    rng = np.random.Generator(np.random.PCG64(2431))
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789 /.-_=&%:", dtype=np.uint8)
    pats = {b"q", b"Z", b"zq", b"0x", b"%%"}
    while len(pats) < 3000:
        u = rng.random()
        ln = int(rng.integers(1, 3)) if u < 0.01 else int(rng.integers(3, 40)) if u < 0.8 else int(rng.integers(40, 244))
        pats.add(alpha[rng.integers(0, alpha.size, ln)].tobytes())
    pats = sorted(pats, key=lambda p: (rng.random(), p))
    data = alpha[rng.integers(0, alpha.size, n)].copy()
pf = wl.write_pattern_file(tempfile.mktemp(), pats)
h = api.PFAC.create()
h.setPerfMode(api.PFAC_SPACE_DRIVEN)
h.setKernelVariant({"naive": api.PFACX_KERNEL_NAIVE, "filter": api.PFACX_KERNEL_FILTER, "auto": api.PFACX_KERNEL_AUTO, "reftable": api.PFACX_KERNEL_REFTABLE}[variant])
h.readPatternFromFile(pf)
d_in = torch.from_numpy(data).to("cuda:0")
d_out = torch.empty(n, dtype=torch.int32, device="cuda:0")
for _ in range(launches):
    h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
torch.cuda.synchronize()
print(case, variant, "matches", int((d_out != 0).sum()), "of", n)
h.destroy()

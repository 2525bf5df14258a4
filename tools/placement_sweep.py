"""tools/placement_sweep.py [workload] -- per-call time of the scan kernel as a function of WHERE the input and
result buffers sit: one 14 GiB slab, the input at slab + a, the results at slab + 6 GiB + b, for a set of
(a, b).  Shows which address bits the two speed classes of DESIGN.md 3.3 follow.  GPU box only."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pfac_amd import api, hiprt, workloads as wl
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = wl.make_config(name); f = tempfile.mktemp(); wl.write_pattern_file(f, cfg.patterns)
h = api.PFAC.create(); h.setPerfMode(cfg.perf_mode); h.setKernelVariant(0); h.readPatternFromFile(f)
n = 1 << 30
host = torch.from_numpy(cfg.input_slice(n + 64, 0))
slab = torch.empty(14 << 30, dtype=torch.uint8, device="cuda:0")
base = slab.data_ptr()
print("slab base", hex(base), "mod 1GiB", hex(base & ((1 << 30) - 1)))
def timeit(a, b):
    d_in = slab[a:a + n + 64]; d_in.copy_(host)
    out_ptr = base + (6 << 30) + b
    for _ in range(30): h.matchFromDevice(d_in.data_ptr(), n, out_ptr)
    e0, e1 = hiprt.Event(), hiprt.Event()
    torch.cuda.synchronize(); e0.record(0)
    for _ in range(20): h.matchFromDevice(d_in.data_ptr(), n, out_ptr)
    e1.record(0); torch.cuda.synchronize()
    return e0.elapsed_ms(e1) / 20
offs = [0, 1 << 12, 1 << 16, 1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 3 << 23, 5 << 21]
print("in-offset sweep (b = 0):")
for a in offs: print("  a=%#11x  %.4f ms" % (a, timeit(a, 0)))
print("out-offset sweep (a = 0):")
for b in offs: print("  b=%#11x  %.4f ms" % (b, timeit(0, b)))
print("both:")
for a in (1 << 23, 1 << 24, 3 << 23):
    for b in (1 << 23, 1 << 24, 3 << 23): print("  a=%#x b=%#x  %.4f ms" % (a, b, timeit(a, b)))
print("repeat a=0 b=0:", [round(timeit(0, 0), 4) for _ in range(3)])

"""tools/placement_tables.py -- same input/result buffers, the handle (= the table allocations) re-created several
times, and the buffers re-allocated several times with one handle: which of the two moves the per-call time
between the two speed classes of DESIGN.md 3.3?  GPU box only."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pfac_amd import api, hiprt, workloads as wl
cfg = wl.make_config("c3"); f = tempfile.mktemp(); wl.write_pattern_file(f, cfg.patterns)
n = 1 << 30
host = torch.from_numpy(cfg.input_slice(n + 64, 0))
def mk():
    h = api.PFAC.create(); h.setPerfMode(cfg.perf_mode); h.setKernelVariant(0); h.readPatternFromFile(f); return h
def timeit(h, d_in, d_out):
    for _ in range(30): h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
    e0, e1 = hiprt.Event(), hiprt.Event()
    torch.cuda.synchronize(); e0.record(0)
    for _ in range(20): h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
    e1.record(0); torch.cuda.synchronize()
    return round(e0.elapsed_ms(e1) / 20, 4)
d_in = host.to("cuda:0"); d_out = torch.empty(n + 64, dtype=torch.int32, device="cuda:0")
print("buffers", hex(d_in.data_ptr()), hex(d_out.data_ptr()))
hs = []
for k in range(6):
    junk = torch.empty((k * 7 + 1) << 20, dtype=torch.uint8, device="cuda:0")   # shift what the next hipMalloc returns
    h = mk(); hs.append((h, junk)); print("handle", k, timeit(h, d_in, d_out))
print("first handle again", timeit(hs[0][0], d_in, d_out))
h = hs[0][0]
keep = []
for k in range(5):
    a = host.to("cuda:0"); b = torch.empty(n + 64, dtype=torch.int32, device="cuda:0"); keep.append((a, b))
    print("buffers", k, hex(a.data_ptr()), hex(b.data_ptr()), timeit(h, a, b))

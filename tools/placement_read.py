"""tools/placement_read.py -- is an input allocation that puts the scan in the slow class also
slower for a plain streaming read and for a plain copy?  GPU box only."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pfac_amd import api, hiprt, workloads as wl
cfg = wl.make_config("c3"); f = tempfile.mktemp(); wl.write_pattern_file(f, cfg.patterns)
n = 1 << 30
host = torch.from_numpy(cfg.input_slice(n + 64, 0))
h = api.PFAC.create(); h.setPerfMode(cfg.perf_mode); h.setKernelVariant(0); h.readPatternFromFile(f)
d_out = torch.empty(n + 64, dtype=torch.int32, device="cuda:0")
def ev(fn, reps):
    for _ in range(5): fn()
    e0, e1 = hiprt.Event(), hiprt.Event()
    torch.cuda.synchronize(); e0.record(0)
    for _ in range(reps): fn()
    e1.record(0); torch.cuda.synchronize()
    return round(e0.elapsed_ms(e1) / reps, 4)
ins = [host.to("cuda:0") for _ in range(6)]
dst = torch.empty(n, dtype=torch.uint8, device="cuda:0")
for k, t in enumerate(ins):
    v = t[:n].view(torch.int64)
    scan = ev(lambda: h.matchFromDevice(t.data_ptr(), n, d_out.data_ptr()), 15)
    red = ev(lambda: torch.sum(v), 10)
    cp = ev(lambda: dst.copy_(t[:n]), 10)
    print(k, hex(t.data_ptr()), "scan", scan, "ms  sum", red, "ms  copy", cp, "ms")

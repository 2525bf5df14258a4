"""tools/placement_probe.py -- same kernel, same input, four different (input, result) buffer pairs in one
process: on some boxes of the pool the per-call time flips between two values with bits 23/24 of the
buffer addresses (1.21 vs 1.30 ms on C3), on others it does not.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, tempfile
from pfac_amd import api, hiprt, workloads as wl
cfg = wl.make_config("c3"); f = tempfile.mktemp(); wl.write_pattern_file(f, cfg.patterns)
h = api.PFAC.create(); h.setPerfMode(cfg.perf_mode); h.setKernelVariant(0); h.readPatternFromFile(f)
n = 1 << 30
host = cfg.input_slice(n + 64, 0)
def timeit(d_in, d_out):
    for _ in range(40): h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
    a, b = hiprt.Event(), hiprt.Event()
    torch.cuda.synchronize(); a.record(0)
    for _ in range(20): h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
    b.record(0); torch.cuda.synchronize()
    return a.elapsed_ms(b) / 20
res = []
keep = []
for k in range(4):
    d_in = torch.from_numpy(host.copy()).to("cuda:0"); d_out = torch.empty(n + 64, dtype=torch.int32, device="cuda:0")
    keep.append((d_in, d_out))
    res.append((hex(d_in.data_ptr()), hex(d_out.data_ptr()), round(timeit(d_in, d_out), 4)))
print(os.getpid(), res)

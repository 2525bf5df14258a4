"""tools/placement_mix.py -- four input buffers x four result buffers, every combination timed: does the slow
speed class follow the input allocation, the result allocation, or the pair?  GPU box only."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pfac_amd import api, hiprt, workloads as wl
cfg = wl.make_config("c3"); f = tempfile.mktemp(); wl.write_pattern_file(f, cfg.patterns)
n = 1 << 30
host = torch.from_numpy(cfg.input_slice(n + 64, 0))
h = api.PFAC.create(); h.setPerfMode(cfg.perf_mode); h.setKernelVariant(0); h.readPatternFromFile(f)
def timeit(d_in, d_out):
    for _ in range(25): h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
    e0, e1 = hiprt.Event(), hiprt.Event()
    torch.cuda.synchronize(); e0.record(0)
    for _ in range(15): h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
    e1.record(0); torch.cuda.synchronize()
    return round(e0.elapsed_ms(e1) / 15, 3)
ins, outs = [], []
for k in range(4):
    ins.append(host.to("cuda:0")); outs.append(torch.empty(n + 64, dtype=torch.int32, device="cuda:0"))
print("in :", [hex(t.data_ptr()) for t in ins]); print("out:", [hex(t.data_ptr()) for t in outs])
for i, a in enumerate(ins): print("in", i, [timeit(a, b) for b in outs])

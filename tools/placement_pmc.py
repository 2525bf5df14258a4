"""tools/placement_pmc.py -- 2 inputs x 4 result buffers, 6 launches per pair, in a fixed order: run under
`rocprofv3 --pmc ... --kernel-trace` and group the per-dispatch counters by pair (tools/placement_pmc_report.py) to
see WHAT differs between a fast and a slow pair of allocations.  GPU box only."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pfac_amd import api, workloads as wl
cfg = wl.make_config("c3"); f = tempfile.mktemp(); wl.write_pattern_file(f, cfg.patterns)
n = 1 << 30
host = torch.from_numpy(cfg.input_slice(n + 64, 0))
h = api.PFAC.create(); h.setPerfMode(cfg.perf_mode); h.setKernelVariant(0); h.readPatternFromFile(f)
ins = [host.to("cuda:0") for _ in range(2)]
outs = [torch.empty(n + 64, dtype=torch.int32, device="cuda:0") for _ in range(4)]
for _ in range(10): h.matchFromDevice(ins[0].data_ptr(), n, outs[0].data_ptr())
torch.cuda.synchronize()
order = []
for i, a in enumerate(ins):
    for j, b in enumerate(outs):
        for _ in range(6):
            h.matchFromDevice(a.data_ptr(), n, b.data_ptr()); order.append((i, j))
torch.cuda.synchronize()
print("ORDER", len(order))

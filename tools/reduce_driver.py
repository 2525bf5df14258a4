"""tools/reduce_driver.py [workload] [calls] [MiB]   (GPU box only)
A few PFAC_matchFromDeviceReduce calls over a BASELINE stream, for profiling: put `python3 tools/reduce_driver.py ...`
directly behind `rocprofv3 ... --` (tools/pmc_run.py --kernel pfac_scan_filter -- tools/reduce_driver.py c3 4)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfac_amd import api, workloads as wl

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = (int(sys.argv[3]) if len(sys.argv) > 3 else 1024) << 20
cfg = wl.make_config(name)
pf = wl.write_pattern_file(tempfile.mktemp(), cfg.patterns)
h = api.PFAC.create()
h.setPerfMode(cfg.perf_mode)
h.readPatternFromFile(pf)
d_in = torch.from_numpy(cfg.input_slice(n, 0)).to("cuda:0")
d_res = torch.empty(n, dtype=torch.int32, device="cuda:0")
d_pos = torch.empty(n, dtype=torch.int32, device="cuda:0")
h.setKernelTiming(True)
for _ in range(calls):
    _, count = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_res.data_ptr(), d_pos.data_ptr())
torch.cuda.synchronize()
st = h.scanStats(n)
print(name, "pairs", count, "kernel ms", st.get("filterKernelMs"), {k: st[k] for k in ("walkerRounds", "laneSteps", "walksStarted", "level1Hits", "ladderCandidates", "walksPerLane") if k in st})
h.destroy()

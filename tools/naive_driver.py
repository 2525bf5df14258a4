"""tools/naive_driver.py [workload] [launches] [variant]   (GPU box only)
A few PFAC_matchFromDevice launches of one kernel variant over a 1 GiB BASELINE stream, for `tools/pmc_run.py --kernel pfac_scan_tiled -- tools/naive_driver.py c3`."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfac_amd import api, workloads as wl
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
variant = {"naive": api.PFACX_KERNEL_NAIVE, "filter": api.PFACX_KERNEL_FILTER, "reftable": api.PFACX_KERNEL_REFTABLE}[sys.argv[3] if len(sys.argv) > 3 else "naive"]
cfg = wl.make_config(name)
pf = wl.write_pattern_file(tempfile.mktemp(), cfg.patterns)
h = api.PFAC.create(); h.setPerfMode(cfg.perf_mode); h.setKernelVariant(variant); h.readPatternFromFile(pf)
n = 1 << 30
d_in = torch.from_numpy(cfg.input_slice(n, 0)).to("cuda:0")
d_out = torch.empty(n, dtype=torch.int32, device="cuda:0")
for _ in range(launches):
    h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
torch.cuda.synchronize()
h.destroy()

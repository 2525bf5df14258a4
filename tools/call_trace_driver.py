"""tools/call_trace_driver.py <MiB> [auto|filter|naive] [calls]   (GPU box only; behind `rocprofv3 --kernel-trace ... --`: tools/call_trace.sh)
PFAC_matchFromDevice calls of one size over the config-3 stream: what a call launches."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfac_amd import api, workloads as wl

n = int(float(sys.argv[1]) * (1 << 20))
variant = {"auto": api.PFACX_KERNEL_AUTO, "filter": api.PFACX_KERNEL_FILTER, "naive": api.PFACX_KERNEL_NAIVE}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cfg = wl.make_config("c3")
pf = wl.write_pattern_file(tempfile.mktemp(), cfg.patterns)
h = api.PFAC.create()
h.setPerfMode(cfg.perf_mode)
h.readPatternFromFile(pf)
h.setKernelVariant(variant)
d_in = torch.from_numpy(cfg.input_slice(n, 0).copy()).to("cuda:0")
d_out = torch.empty(n, dtype=torch.int32, device="cuda:0")
for _ in range(calls):
    h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
torch.cuda.synchronize()
h.destroy()

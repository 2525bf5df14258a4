"""tools/call_breakdown.py [workload] -- PFAC_matchFromDevice on the 1 GiB bench stream: GPU time of the whole call (HIP
events around it, calls back to back) next to the filter kernel alone (PFACX_setKernelTiming): what the counters'
memset in front of the kernel and the simple kernel behind it (end of the input, dense chunks) cost per call."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfac_amd import api, hiprt, workloads as wl

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = wl.make_config(name)
pf = wl.write_pattern_file(f"/tmp/breakdown_{name}.pat", cfg.patterns)
n = 1 << 30
h = api.PFAC.create()
h.setPerfMode(api.PFAC_SPACE_DRIVEN if cfg.perf_mode else api.PFAC_TIME_DRIVEN)
h.readPatternFromFile(pf)
d_in = torch.from_numpy(cfg.input_slice(n, 0)).to("cuda:0")
d_out = torch.empty(n, dtype=torch.int32, device="cuda:0")
for _ in range(3):
    h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
torch.cuda.synchronize()
for timing in (False, True):
    h.setKernelTiming(timing)
    ev = [(hiprt.Event(), hiprt.Event()) for _ in range(20)]
    kern = []
    for a, b in ev:
        a.record(0)
        h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
        b.record(0)
        if timing:
            kern.append(h.scanStats()["filterKernelMs"])          # waits for the stream
    torch.cuda.synchronize()
    call = [a.elapsed_ms(b) for a, b in ev]
    print(f"{name} kernel timing {'on ' if timing else 'off'}: call median {np.median(call):.4f} ms min {min(call):.4f}" +
          (f" | filter kernel alone median {np.median(kern):.4f} min {min(kern):.4f} | around it {np.median(call) - np.median(kern):.4f} ms" if timing else ""))
# whole batch, no events between the calls
a, b = hiprt.Event(), hiprt.Event()
h.setKernelTiming(False)
a.record(0)
for _ in range(20):
    h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
b.record(0)
torch.cuda.synchronize()
print(f"{name} 20 calls back to back: {a.elapsed_ms(b) / 20:.4f} ms per call")
h.destroy()

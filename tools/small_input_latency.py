"""tools/small_input_latency.py -- per-call time of PFAC_matchFromDevice for small inputs (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time, tempfile
from pfac_amd import api, workloads as wl
cfg = wl.make_config("c3"); f = tempfile.mktemp(); wl.write_pattern_file(f, cfg.patterns)
h = api.PFAC.create(); h.setPerfMode(cfg.perf_mode); h.readPatternFromFile(f)
for variant, vname in ((api.PFACX_KERNEL_AUTO, "auto"), (api.PFACX_KERNEL_FILTER, "filter"), (api.PFACX_KERNEL_NAIVE, "naive")):
  h.setKernelVariant(variant)
  for n in [int(x) << 10 for x in os.environ.get('PFAC_LAT_KIB', '4,64,256,512,768,1024,4096,16384').split(',')]:
      d_in = torch.from_numpy(cfg.input_slice(n, 0).copy()).to("cuda:0"); d_out = torch.empty(n, dtype=torch.int32, device="cuda:0")
      for _ in range(5): h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
      torch.cuda.synchronize()
      reps = 200 if n <= (16 << 20) else 20
      t0 = time.perf_counter()
      for _ in range(reps): h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
      t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
      print(vname, "n=%9d: %.1f us per call (host-side enqueue %.1f us), %.2f GB/s" % (n, (t2 - t0) / reps * 1e6, (t1 - t0) / reps * 1e6, n * reps / (t2 - t0) / 1e9))

"""tools/host_path.py [MiB]  -- PFAC_matchFromHost on the Snort-style stream, pageable and pinned buffers, against the link (GPU box only)"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfac_amd import api, workloads as wl
hn = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
cfg = wl.make_config("c3"); pf = wl.write_pattern_file(tempfile.mktemp(), cfg.patterns)
h = api.PFAC.create(); h.setPerfMode(cfg.perf_mode); h.readPatternFromFile(pf)
host = cfg.input_slice(hn, 0).copy()
d_in = torch.from_numpy(host).to("cuda:0"); d_out = torch.empty(hn, dtype=torch.int32, device="cuda:0")
h.matchFromDevice(d_in.data_ptr(), hn, d_out.data_ptr()); torch.cuda.synchronize()
want = d_out.cpu().numpy()
for kind in ("pageable", "pinned"):
    h_in = torch.from_numpy(host.copy()); h_out = torch.empty(hn, dtype=torch.int32)
    if kind == "pinned": h_in, h_out = h_in.pin_memory(), h_out.pin_memory()
    h.matchFromHost(h_in.data_ptr(), hn, h_out.data_ptr())
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); h.matchFromHost(h_in.data_ptr(), hn, h_out.data_ptr()); ts.append(time.perf_counter() - t0)
    print(kind, "best %.3f ms median %.3f ms -> %.1f GB/s (median), same result %s" % (min(ts) * 1e3, sorted(ts)[2] * 1e3, hn / sorted(ts)[2] / 1e9, bool(np.array_equal(h_out.numpy(), want))))
probe = torch.from_numpy(host).pin_memory(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): d_in.copy_(probe, non_blocking=True)
torch.cuda.synchronize(); print("link h2d pinned %.1f GB/s" % (3 * hn / (time.perf_counter() - t0) / 1e9), "cores allowed", len(os.sched_getaffinity(0)))

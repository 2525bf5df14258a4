"""tools/stress_calls.py [calls] [seed] [workload = c3 | c6] -- one handle, many calls of changing shape: random slices (any alignment, 1 byte ..
48 MiB) of one 64 MiB Snort-style buffer through PFAC_matchFromDevice and PFAC_matchFromDeviceReduce in random order, every
result against the reference-shaped kernel's (a second handle with PFACX_KERNEL_REFTABLE: the independent implementation) on the same
slice.  What it is after: state that one launch leaves for the next (launch counters left zero by the last block out,
the two dense-chunk counters, the ordering scratch)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfac_amd import api, workloads as wl

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 7))
cfg = wl.make_config(sys.argv[3] if len(sys.argv) > 3 else "c3")       # c6: near misses over the Snort-scale set (PFAC_TEST_WALKER=veto: the VETO = 2 kernel on every big slice)
pf = wl.write_pattern_file("/tmp/stress_calls.pat", cfg.patterns)
N = 64 << 20
host = cfg.input_slice(N, 0).copy()
host[5 << 20:(5 << 20) + (3 << 20)] = ord("a")              # a pattern-dense stretch (if "a..." hits level 1) and a run of one byte
d_buf = torch.from_numpy(host).to("cuda:0")


def handle(variant):
    h = api.PFAC.create()
    h.setPerfMode(api.PFAC_SPACE_DRIVEN)
    h.readPatternFromFile(pf)
    h.setKernelVariant(variant)
    return h


a, b = handle(api.PFACX_KERNEL_AUTO), handle(api.PFACX_KERNEL_REFTABLE)
d_out = torch.empty(N + 16, dtype=torch.int32, device="cuda:0")
d_ref = torch.empty(N + 16, dtype=torch.int32, device="cuda:0")
d_res = torch.empty(N, dtype=torch.int32, device="cuda:0")
d_pos = torch.empty(N, dtype=torch.int32, device="cuda:0")
bad = 0
for k in range(calls):
    kind = int(rng.integers(0, 4))
    n = int(rng.integers(1, 4096)) if kind == 0 else int(rng.integers(1 << 20, 3 << 20)) if kind == 1 else int(rng.integers(1 << 20, 48 << 20))
    off = int(rng.integers(0, N - n + 1))
    if rng.random() < 0.5:
        off &= ~15
    out_off = int(rng.integers(0, 4))
    b.matchFromDevice(d_buf.data_ptr() + off, n, d_ref.data_ptr())
    if rng.random() < 0.5:
        d_out.fill_(-7)
        a.matchFromDevice(d_buf.data_ptr() + off, n, d_out.data_ptr() + 4 * out_off)
        ok = torch.equal(d_out[out_off:out_off + n], d_ref[:n]) and int(d_out[out_off + n]) == -7
        what = "full"
    else:
        _, count = a.matchFromDeviceReduce(d_buf.data_ptr() + off, n, d_res.data_ptr(), d_pos.data_ptr())
        nz = torch.nonzero(d_ref[:n]).flatten()
        ok = count == nz.numel() and torch.equal(d_pos[:count].to(torch.int64), nz) and torch.equal(d_res[:count], d_ref[:n][nz])
        what = "reduce"
    if not ok:
        bad += 1
        print(f"MISMATCH call {k}: {what} off {off} n {n} out_off {out_off}", flush=True)
    if k % 50 == 49:
        print(f"{k + 1} calls, mismatches so far {bad}", flush=True)
a.destroy(); b.destroy()
print("stress_calls done, mismatches:", bad)
sys.exit(1 if bad else 0)

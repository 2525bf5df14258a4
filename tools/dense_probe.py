import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pfac_amd import api, workloads as wl
rng = np.random.Generator(np.random.PCG64(2431))
alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789 /.-_=&%:", dtype=np.uint8)
pats = {b"q", b"Z", b"zq", b"0x", b"%%"}
while len(pats) < 3000:
    u = rng.random()
    ln = int(rng.integers(1, 3)) if u < 0.01 else int(rng.integers(3, 40)) if u < 0.8 else int(rng.integers(40, 244))
    pats.add(alpha[rng.integers(0, alpha.size, ln)].tobytes())
pats = sorted(pats)
pf = wl.write_pattern_file("/tmp/snortlen.pat", pats)
n = 64 << 20
data = alpha[rng.integers(0, alpha.size, n)].copy()
d_in = torch.from_numpy(data).to("cuda:0"); d_out = torch.empty(n, dtype=torch.int32, device="cuda:0")
h = api.PFAC.create(); h.setPerfMode(1); h.setTextureMode(1); h.setKernelVariant(api.PFACX_KERNEL_AUTO); h.readPatternFromFile(pf)
for call in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr()); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    i = h.info(); st = h.scanStats()
    print(call, "%.3f ms %.1f GB/s" % (dt * 1e3, n / dt / 1e9), "streamDense", i.streamDense, "nearMiss", i.streamNearMisses, "denseChunks", st["denseChunks"], "l1hits", st["level1Hits"])

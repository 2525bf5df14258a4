"""tools/stress_fuzz.py [first_seed] [count] -- more seeds of tests/test_full_result.py::test_fuzzed_pattern_sets_over_tiny_alphabets
(pattern sets over tiny alphabets, prefixes at every depth, 1-2 byte patterns), bigger inputs, all four table modes,
full-result and compacted-output calls, each launch repeated: a soak run for the GPU box, not part of the suite."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import binding as ob
from pfac_amd import api, workloads as wl

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
MODES = [(api.PFAC_TIME_DRIVEN, api.PFAC_TEXTURE_OFF), (api.PFAC_TIME_DRIVEN, api.PFAC_TEXTURE_ON),
         (api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_OFF), (api.PFAC_SPACE_DRIVEN, api.PFAC_TEXTURE_ON)]
work = tempfile.mkdtemp()
bad = 0
for seed in range(first, first + count):
    rng = np.random.Generator(np.random.PCG64(seed))
    alphabet = [bytes([b]) for b in rng.choice([0x00, 0xFF, 0x41, 0x42, 0x7A, 0x20, 0x0D, 0x61, 0x2F], size=int(rng.integers(2, 7)), replace=False)]
    pats = set()
    base = b"".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), 64))
    for cut in rng.integers(1, 64, int(rng.integers(3, 20))):
        pats.add(base[:int(cut)])
    while len(pats) < int(rng.integers(8, 400)):
        ln = int(rng.integers(1 if seed % 2 else 3, 70))
        pats.add(b"".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), ln)))
    pats = sorted(pats, key=lambda p: (rng.random(), p))
    pf = wl.write_pattern_file(os.path.join(work, f"s{seed}.pat"), pats)
    n = int(rng.integers(1 << 20, 6 << 20))
    skew = rng.random(len(alphabet)) ** 3 + 0.01
    idx = rng.choice(len(alphabet), size=n, p=skew / skew.sum())
    data = np.frombuffer(b"".join(alphabet), dtype=np.uint8)[idx].copy()
    for _ in range(20):
        at = int(rng.integers(0, n - 100))
        data[at:at + len(base)] = np.frombuffer(base, dtype=np.uint8)
    o = ob.Oracle(pf, hashed=False)
    want = o.match(data, omp=True)
    o.close()
    nz = np.nonzero(want)[0]
    d_in = torch.from_numpy(data).to("cuda:0")
    for perf, tex in MODES:
        h = api.PFAC.create()
        h.setPerfMode(perf); h.setTextureMode(tex); h.readPatternFromFile(pf)
        try:
            for variant, reps in ((api.PFACX_KERNEL_NAIVE, 2), (api.PFACX_KERNEL_FILTER, 3)):       # the tiled kernel alone, then the filter kernel (+ its dense chunks through the tiled one)
                h.setKernelVariant(variant)
                for rep in range(reps):
                    d_out = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
                    h.matchFromDevice(d_in.data_ptr(), n, d_out.data_ptr())
                    got = d_out.cpu().numpy()
                    if not np.array_equal(got, want):
                        w = np.nonzero(got != want)[0]
                        print(f"MISMATCH seed {seed} mode {perf}/{tex} variant {variant} rep {rep}: {w.size} positions, first {w[0]} got {got[w[0]]} want {want[w[0]]}", flush=True)
                        bad += 1
                if variant == api.PFACX_KERNEL_NAIVE:                                                 # compacted output through the tiled kernel too
                    d_res = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
                    d_pos = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
                    st, cnt = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_res.data_ptr(), d_pos.data_ptr())
                    if not (cnt == nz.size and np.array_equal(d_pos[:cnt].cpu().numpy(), nz) and np.array_equal(d_res[:cnt].cpu().numpy(), want[nz])):
                        print(f"REDUCE MISMATCH (tiled) seed {seed} mode {perf}/{tex}", flush=True)
                        bad += 1
            d_res = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
            d_pos = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
            st, cnt = h.matchFromDeviceReduce(d_in.data_ptr(), n, d_res.data_ptr(), d_pos.data_ptr())
            if not (cnt == nz.size and np.array_equal(d_pos[:cnt].cpu().numpy(), nz) and np.array_equal(d_res[:cnt].cpu().numpy(), want[nz])):
                print(f"REDUCE MISMATCH seed {seed} mode {perf}/{tex}", flush=True)
                bad += 1
            host = np.full(n, -7, dtype=np.int32)
            h.matchFromHost(data.ctypes.data, n, host.ctypes.data)
            if not np.array_equal(host, want):
                print(f"HOST MISMATCH seed {seed} mode {perf}/{tex}", flush=True)
                bad += 1
        finally:
            h.destroy()
    if (seed - first) % 10 == 9:
        print(f"seed {seed}: {len(pats)} patterns, {n} bytes, {nz.size} matches; mismatches so far {bad}", flush=True)
print("stress done, mismatches:", bad)
sys.exit(1 if bad else 0)

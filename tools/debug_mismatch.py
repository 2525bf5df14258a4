import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfac_amd import api, workloads as wl
from oracle import binding as ob
tmp = 'gpurun_out/dbg'; os.makedirs(tmp, exist_ok=True)
p2 = wl.random_patterns()
d2 = wl.random_bytes((1 << 20) + 37).copy()
rng = np.random.Generator(np.random.PCG64(11))
for p in p2[:200]:
    at = int(rng.integers(0, d2.size - 64)); d2[at:at+len(p)] = np.frombuffer(p, dtype=np.uint8)
pf = wl.write_pattern_file(os.path.join(tmp, 'c2.pat'), p2)
o = ob.Oracle(pf, hashed=False); want = o.match(d2)
for variant in (0, 1):
    h = api.PFAC.create(); h.setTextureMode(api.PFAC_TEXTURE_OFF); h.setKernelVariant(variant); h.readPatternFromFile(pf)
    d_in = torch.from_numpy(d2).cuda(); d_out = torch.full((d2.size,), -5, dtype=torch.int32, device='cuda')
    for rep in range(3):
        d_out.fill_(-5)
        h.matchFromDevice(d_in.data_ptr(), d2.size, d_out.data_ptr()); torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        bad = np.nonzero(got != want)[0]
        print('variant', variant, 'rep', rep, 'mismatches', bad.size, 'nonzero want', int((want != 0).sum()), 'poison left', int((got == -5).sum()))
    for b in bad[:40]:
        pid = want[b]; t = b % 1024
        print(f"  pos {b} tile {b//1024} in-tile {t} k {t//256} lane {(t%256)//4} i {t%4} got {got[b]} want {pid} len {len(p2[pid-1]) if pid>0 else 0}")
    h.destroy()

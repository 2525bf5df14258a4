"""tools/placement_probe.py [GiB of one allocation]   (GPU box only)
The bare 1R:4W stream (pfac_stream_1r4w of the kernel module: every wave reads 1 KiB and writes 4 KiB of zeros) over 1 GiB of input at offset A and
4 GiB of output at offset B of ONE device allocation, for a grid of (A, B), and over separately allocated buffers: is the placement class of a
process's buffers (profiles/r06_c6_placement_classes.txt) a matter of where the two streams lie relative to each other, or of the allocation?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfac_amd import api

G = 1 << 30
total = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mod = C.CDLL(api.library_paths()[1])
mod.PFACX_streamProbe.restype = C.c_double
mod.PFACX_streamProbe.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
big = torch.empty(total * G, dtype=torch.uint8, device="cuda:0")
base = big.data_ptr()
print("one allocation of", total, "GiB at", hex(base))
def probe(a, b):
    return float(mod.PFACX_streamProbe(base + a, base + b, G, 10))
probe(0, 2 * G)
q = G // 4
print("rows: input offset A; columns: output offset B (GiB) ->", " ".join("%5.2f" % (b / G) for b in range(2 * G, (total - 4) * G + 1, 3 * q)))
for a in (0, q, 2 * q, 3 * q, G):
    print("A %.2f:" % (a / G), " ".join("%.3f" % probe(a, b) for b in range(2 * G, (total - 4) * G + 1, 3 * q)))
print("fine steps of B from 2 GiB (A = 0):", " ".join("%d KiB %.3f" % (d >> 10, probe(0, 2 * G + d)) for d in (0, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, 64 << 20, 128 << 20)))
del big
torch.cuda.empty_cache()
pairs = []
for k in range(4):
    i = torch.empty(G, dtype=torch.uint8, device="cuda:0")
    o = torch.empty(4 * G, dtype=torch.uint8, device="cuda:0")
    pairs.append((i, o))
print("separately allocated pairs:", " ".join("%s/%s %.3f" % (hex(i.data_ptr()), hex(o.data_ptr()), float(mod.PFACX_streamProbe(i.data_ptr(), o.data_ptr(), G, 10))) for i, o in pairs))
print("crossed (input of pair k, output of pair k+1):", " ".join("%.3f" % float(mod.PFACX_streamProbe(pairs[k][0].data_ptr(), pairs[(k + 1) % 4][1].data_ptr(), G, 10)) for k in range(4)))

"""tools/chain_steps.py [c5|c3|c2]  -- size of the chained table and the table steps a walk takes through it on a sample
of the workload's stream (host-only handle, python walker: the model of tests/test_host_api.py).  CPU only."""
import sys
import numpy as np
sys.path.insert(0, ".")
from pfac_amd import api, workloads as wl
import tempfile, os

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
cfg = wl.make_config(name)
d = tempfile.mkdtemp()
pf = wl.write_pattern_file(os.path.join(d, "p.pat"), cfg.patterns)
h = api.PFAC.createHostOnly()
h.setPerfMode(cfg.perf_mode)
h.readPatternFromFile(pf)
slots = h.table(api.PFACX_TABLE_CHAIN).reshape(-1, 4)
info = h.info()
J = info.chainJumpLog2
N = len(slots) // 2                     # slot headers; as many extension units behind them
LONG = len(sys.argv) > 2 and sys.argv[2] == 'long'
jump_base = N - (2 << J) + ((1 << J) if LONG else 0)
root_row = N - (2 << J) - 256
print(f"{name}: states {info.numOfStates} chain slots {len(slots)} ({len(slots) * 16 / 1e6:.2f} MB), buckets {root_row * 16 / 1e6:.2f} MB, jump 2^{J}")
data = bytes(cfg.input_slice(1 << 20)) + bytes(128)
EMPTY, FINAL, WIDE = 1 << 14, 1 << 13, 1 << 15

def step(at, ext, b0, p):
    s = slots[at]; meta = int(s[0]); ln = (meta >> 8) & 0x1F
    chain = int(s[2]).to_bytes(4, "little") + int(s[3]).to_bytes(4, "little")
    if ln > 7: chain += b"".join(int(v).to_bytes(4, "little") for v in slots[at + N])
    ok = (meta & (EMPTY | 0xFF)) == b0 and chain[:ln] == data[p:p + ln]
    return ok, (meta >> 16) & 0xFF == 0, int(s[1]), meta, 1 + ln

hist = {}
walks = steps = gathers = 0
for i in range(0, (1 << 20) - 200):
    x = int.from_bytes(data[i:i + 4], "little")
    n = 1; g = 1
    ok, leaf, row, ks, used = step(jump_base + (((x * 0x9E3779B1) & 0xFFFFFFFF) >> (32 - J)), 0, data[i], i + 1)
    if LONG: g = 2
    if not ok:
        continue                     # only walks the jump table knows (what the prefilter lets through, roughly)
    depth = 0
    while ok:
        depth += used
        if leaf: break
        b0 = data[i + depth]
        r = ((((ks >> 16) & 0xFF) * b0) >> 7) & (ks >> 24)
        wide = bool(ks & WIDE)
        ok, leaf, row, ks, used = step(row + r, (ks >> 24) + 1, b0, i + depth + 1)
        n += 1; g += 2 if wide else 1
    walks += 1; steps += n; gathers += g
    hist[n] = hist.get(n, 0) + 1
print(f"walks from a jump slot: {walks} per MiB, {steps / max(walks, 1):.2f} table steps, {gathers / max(walks, 1):.2f} slot-unit gathers per walk")
print("steps histogram:", dict(sorted(hist.items())))

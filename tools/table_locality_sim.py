"""tools/table_locality_sim.py -- CPU model (development only): cache lines touched by the walkers' gathered slot loads on
the C3 stream, for the table laid out in the reference's state order (states of one pattern adjacent) and in
breadth-first order (shallow states adjacent).  LRU of N 128-byte lines."""
import sys, collections
sys.path.insert(0, '/root/repo')
from pfac_amd import workloads as wl
name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
cfg = wl.make_config(name)
pats = sorted(cfg.patterns)
nxt = [{}]; final = [False]; depth = [0]
for p in pats:
    s = 0
    for ch in p:
        t = nxt[s].get(ch)
        if t is None:
            t = len(nxt); nxt.append({}); final.append(False); depth.append(depth[s] + 1); nxt[s][ch] = t
        s = t
    final[s] = True
N = len(nxt)
def pow2(x):
    p = 1
    while p < x: p *= 2
    return p
size = [pow2(len(nxt[s])) if nxt[s] else 0 for s in range(N)]
def layout(order):
    row = [0] * N; at = 0
    for s in order:
        row[s] = at; at += size[s]
    return row, at
orders = {"reference order (insertion of sorted patterns)": list(range(N)),
          "breadth-first": sorted(range(N), key=lambda s: (depth[s], s))}
data = cfg.input_slice(4 << 20, 0).tobytes()
n = len(data) - 80
# the gathers of every walk: (state whose bucket is probed, byte)
def walks():
    for i in range(n):
        s = nxt[0].get(data[i])
        if s is None: continue
        d = 1; t = s; ok = False
        while d < 4:
            if final[t]: ok = True; break
            t2 = nxt[t].get(data[i + d])
            if t2 is None: break
            t = t2; d += 1
        else: ok = True
        if not ok: continue
        d = 0; s = 0; first = True; probes = []
        while True:
            t = nxt[s].get(data[i + d])
            if not first: probes.append((s, data[i + d]))
            if t is None: break
            d += 1; s = t; first = False
            k = 0; dead = False
            while k < 7 and not final[s] and len(nxt[s]) == 1:
                (ch, t2), = nxt[s].items()
                if data[i + d] != ch: dead = True; break
                s = t2; d += 1; k += 1
            if dead or not nxt[s]: break
        yield probes
allp = list(walks())
tot = sum(len(p) for p in allp)
print(name, "walks", len(allp), "gathers", tot, "states", N)
for label, order in orders.items():
    row, slots = layout(order)
    lines = []
    for probes in allp:
        for s, b in probes:
            lines.append((row[s] + (b * 40503 >> 4) % size[s]) // 8)
    distinct = len(set(lines))
    out = "%-48s slots %d (%.1f MB)  distinct lines %d (%.0f KiB)" % (label, slots, slots * 16 / 1e6, distinct, distinct * 128 / 1024)
    for cap in (64, 256, 1024, 4096, 16384):
        lru = collections.OrderedDict(); miss = 0
        for l in lines:
            if l in lru: lru.move_to_end(l)
            else:
                miss += 1; lru[l] = 1
                if len(lru) > cap: lru.popitem(last=False)
        out += "  LRU%d miss %.3f" % (cap, miss / len(lines))
    print(out)

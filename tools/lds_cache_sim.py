"""tools/lds_cache_sim.py -- CPU model (development only): hit rate of a direct-mapped, slot-granular cache of chained-table
slots (one per block, in LDS) over one block's share of the C3 stream (4 MiB), cold at the start of the launch."""
import sys, collections
sys.path.insert(0, '/root/repo')
from pfac_amd import workloads as wl
cfg = wl.make_config('c3')
pats = sorted(cfg.patterns)
nxt = [{}]; final = [False]
for p in pats:
    s = 0
    for ch in p:
        t = nxt[s].get(ch)
        if t is None:
            t = len(nxt); nxt.append({}); final.append(False); nxt[s][ch] = t
        s = t
    final[s] = True
data = cfg.input_slice(4 << 20, 0).tobytes()
n = len(data) - 80
keys = []          # one key per gathered load: ('j', 4-byte prefix) for the jump step, (state, byte) for the others
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
    # jump step
    t = 0; okj = True
    for k in range(4):
        t = nxt[t].get(data[i + k])
        if t is None or (k < 3 and final[t]): okj = False; break
    keys.append(('j', data[i:i + 4]))
    if okj:
        s = t; d = 4
    else:
        s = 0; d = 0
    first = not okj
    while True:
        if not first or not okj:
            pass
        # chain
        k = 0; dead = False
        while k < (4 if (okj and d == 4 and first is False) else 7) and not final[s] and len(nxt[s]) == 1 and s != 0:
            (ch, t2), = nxt[s].items()
            if data[i + d] != ch: dead = True; break
            s = t2; d += 1; k += 1
        if dead or (s != 0 and not nxt[s]): break
        t = nxt[s].get(data[i + d])
        keys.append((s, data[i + d]))
        if t is None: break
        s = t; d += 1
print("gathered loads", len(keys), "distinct", len(set(keys)))
for S in (256, 512, 1024, 2048, 4096):
    tags = {}
    hit = 0
    for k in keys:
        slot = hash(k) % S
        if tags.get(slot) == k: hit += 1
        else: tags[slot] = k
    print("direct-mapped %5d slots (%3d KiB + tags %2d KiB): hit rate %.3f" % (S, S * 16 // 1024, S * 4 // 1024, hit / len(keys)))

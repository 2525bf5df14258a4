"""tools/walk_depth_sim2.py -- CPU model of walk-shortening ideas on the C3 stream (development only):
gathers per walk for (a) today's root-in-LDS + chained slots, (b) a child filter in the slot that ends a walk
without the failing probe, (c) a first step keyed by the first 4 bytes."""
import sys, collections, time
sys.path.insert(0, '/root/repo')
from pfac_amd import workloads as wl
name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
cfg = wl.make_config(name)
pats = cfg.patterns
nxt = [{}]; final = [False]
for p in pats:
    s = 0
    for ch in p:
        t = nxt[s].get(ch)
        if t is None:
            t = len(nxt); nxt.append({}); final.append(False); nxt[s][ch] = t
        s = t
    final[s] = True
data = cfg.input_slice(2 << 20, 0).tobytes()
n = len(data) - 80

def level2(i):
    s = nxt[0].get(data[i])
    if s is None: return False
    d = 1; t = s
    while d < 4:
        if final[t]: return True
        t2 = nxt[t].get(data[i + d])
        if t2 is None: return False
        t = t2; d += 1
    return True
starts = [i for i in range(n) if level2(i)]
print(name, "walks", len(starts), "of", n, "positions")

def chain_from(s, i, d, cap):
    """follow the single-successor chain from state s (input position i+d); returns (state, depth, dead)"""
    k = 0
    while k < cap and not final[s] and len(nxt[s]) == 1:
        (ch, t2), = nxt[s].items()
        if data[i + d] != ch: return s, d, True
        s = t2; d += 1; k += 1
    return s, d, False

def bloom_bit(b): return (b * 0x9E) >> 3 & 31        # any 5-bit hash of the byte

def sim(cap, rootcap, childfilter=False, jump4=False, jumpcap=0):
    gathers = 0; fails = 0; saved = 0; jmiss = 0
    for i in starts:
        d = 0; s = 0; first = True
        if jump4:
            t = 0; ok = True
            for k in range(4):
                t = nxt[t].get(data[i + k])
                if t is None: ok = False; break
            gathers += 1
            if not ok:
                jmiss += 1          # walk over (a shorter pattern matched or a filter false positive): resolved by the slow path
                continue
            s = t; d = 4
            # chain behind the jump (only if no state on the way 1..4 needs anything: finals are folded into the slot)
            s, d, dead = chain_from(s, i, d, jumpcap)
            if dead or not nxt[s]: continue
            first = False
        while True:
            t = nxt[s].get(data[i + d])
            if not first:
                if t is None:
                    if childfilter:
                        bits = 0
                        for ch in nxt[s]: bits |= 1 << bloom_bit(ch)
                        if not (bits >> bloom_bit(data[i + d])) & 1:
                            saved += 1; break
                    gathers += 1; fails += 1
                    break
                gathers += 1
            elif t is None: break
            d += 1; s = t
            c = rootcap if first else cap
            first = False
            s, d, dead = chain_from(s, i, d, c)
            if dead or not nxt[s]: break
    return gathers / len(starts), fails / len(starts), saved / len(starts), jmiss / len(starts)

for label, kw in (("today cap7", dict(cap=7, rootcap=7)),
                  ("cap4", dict(cap=4, rootcap=7)),
                  ("cap3", dict(cap=3, rootcap=7)),
                  ("cap7 + child filter", dict(cap=7, rootcap=7, childfilter=True)),
                  ("cap4 + child filter", dict(cap=4, rootcap=7, childfilter=True)),
                  ("cap3 + child filter", dict(cap=3, rootcap=7, childfilter=True)),
                  ("jump4 chain0, cap7", dict(cap=7, rootcap=7, jump4=True, jumpcap=0)),
                  ("jump4 chain4, cap7", dict(cap=7, rootcap=7, jump4=True, jumpcap=4)),
                  ("jump4 chain7, cap7", dict(cap=7, rootcap=7, jump4=True, jumpcap=7)),
                  ("jump4 chain4, cap4 + child filter", dict(cap=4, rootcap=7, jump4=True, jumpcap=4, childfilter=True)),
                  ):
    t0 = time.time()
    g, f, sv, jm = sim(**kw)
    print("%-36s gathers/walk %.3f  failing probes %.3f  saved by child filter %.3f  jump misses %.3f  (%.0fs)" % (label, g, f, sv, jm, time.time() - t0))

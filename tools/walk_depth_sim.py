"""tools/walk_depth_sim.py -- CPU model of the chained walk on the C3 stream: table steps per walk for different chain
capacities, and the histogram of input bytes a walk examines (why a queue entry carries 20 bytes).  Development only."""
import sys, numpy as np, collections, time
sys.path.insert(0,'/root/repo')
from pfac_amd import workloads as wl
cfg = wl.make_config('c3')
pats = cfg.patterns
# trie
nxt=[{}]; final=[False]
for p in pats:
    s=0
    for ch in p:
        t=nxt[s].get(ch)
        if t is None:
            t=len(nxt); nxt.append({}); final.append(False); nxt[s][ch]=t
        s=t
    final[s]=True
data = cfg.input_slice(2<<20, 0).tobytes()
n=len(data)-80
def sim(cap, rootcap, finalcut=True):
    steps_hist=collections.Counter(); walks=0; total_steps=0
    for i in range(n):
        s=nxt[0].get(data[i])
        if s is None: continue
        # quick L2-like: survive to depth 4 (or final within 3)
        d=1; t=s; ok=False
        while d<4:
            if final[t]: ok=True
            t2=nxt[t].get(data[i+d])
            if t2 is None: break
            t=t2; d+=1
        else: ok=True
        if not ok: continue
        walks+=1
        # chained walk
        d=0; s=0; steps=0
        while True:
            t=nxt[s].get(data[i+d])
            if t is None: break
            steps+=1; d+=1; s=t
            c=rootcap if steps==1 else cap
            k=0; dead=False
            while k<c and not final[s] and len(nxt[s])==1:
                (ch,t2),=nxt[s].items()
                if data[i+d]!=ch: dead=True; break
                s=t2; d+=1; k+=1
            if dead or not nxt[s]: break
        total_steps+=steps
    return walks, total_steps/walks
t0=time.time()
for cap,rc in ((7,7),(11,7),(11,11),(15,15),(23,23),(7,11),(7,23)):
    w,a=sim(cap,rc); print("cap",cap,"rootcap",rc,"walks",w,"avg steps incl root %.3f -> global %.3f"%(a,a-1), "%.0fs"%(time.time()-t0))
# bytes examined per walk (the byte that kills the walk included)
hist=collections.Counter()
for i in range(n):
    s=nxt[0].get(data[i])
    if s is None: continue
    d=1; t=s; ok=False
    while d<4:
        if final[t]: ok=True
        t2=nxt[t].get(data[i+d])
        if t2 is None: break
        t=t2; d+=1
    else: ok=True
    if not ok: continue
    d=0; s=0
    while True:
        t=nxt[s].get(data[i+d])
        d+=1
        if t is None: break
        s=t
        if not nxt[s]: break
    hist[d]+=1
tot=sum(hist.values()); acc=0
for k in sorted(hist):
    acc+=hist[k]; print(k, hist[k], "%.3f"%(acc/tot))

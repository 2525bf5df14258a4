"""tools/host_numa_probe.py [MiB]  -- PFAC_matchFromHost / ...Reduce from pinned and pageable buffers first-touched on each NUMA node
of the host (GPU box only): where the caller's buffers live against where the GPU hangs.  min / p50 / p90 of 20 calls each;
PFAC_HOST_TRACE=1 prints the phases of every call."""
import glob, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfac_amd import api, workloads as wl

def cpus_of(node):
    out = set()
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out

nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
gpu_nodes = {}
for dev in glob.glob("/sys/class/drm/card*/device"):
    try:
        if open(dev + "/vendor").read().strip() == "0x1002":
            gpu_nodes[os.path.basename(os.path.dirname(dev))] = open(dev + "/numa_node").read().strip()
    except OSError:
        pass
allcpus = os.sched_getaffinity(0)
print("NUMA nodes", nodes, {n: len(cpus_of(n) & allcpus) for n in nodes}, "cpus allowed", len(allcpus), "| GPU numa_node", gpu_nodes)
hn = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
cfg = wl.make_config("c3"); pf = wl.write_pattern_file(tempfile.mktemp(), cfg.patterns)
h = api.PFAC.create(); h.setPerfMode(cfg.perf_mode); h.readPatternFromFile(pf)
host = cfg.input_slice(hn, 0).copy()
def stats(ts):
    ts = sorted(ts)
    return "min %.3f p50 %.3f p90 %.3f max %.3f ms -> %.1f GB/s (p50)" % (ts[0] * 1e3, ts[len(ts) // 2] * 1e3, ts[int(len(ts) * 0.9)] * 1e3, ts[-1] * 1e3, hn / ts[len(ts) // 2] / 1e9)
for node in nodes:
    mine = cpus_of(node) & allcpus
    if not mine:
        continue
    for kind in ("pinned", "pageable"):
        os.sched_setaffinity(0, mine)                       # first touch on this node
        h_in = torch.from_numpy(host.copy()); h_out = torch.zeros(hn, dtype=torch.int32)
        if kind == "pinned":
            h_in, h_out = h_in.pin_memory(), h_out.pin_memory()
        r_ids = torch.zeros(hn, dtype=torch.int32); r_pos = torch.zeros(hn, dtype=torch.int32)
        os.sched_setaffinity(0, allcpus)
        for _ in range(2):
            h.matchFromHost(h_in.data_ptr(), hn, h_out.data_ptr())
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); h.matchFromHost(h_in.data_ptr(), hn, h_out.data_ptr()); ts.append(time.perf_counter() - t0)
        print(f"buffers on node {node}, {kind:8s} matchFromHost       :", stats(ts))
        for _ in range(2):
            h.matchFromHostReduce(h_in.data_ptr(), hn, r_ids.data_ptr(), r_pos.data_ptr())
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); h.matchFromHostReduce(h_in.data_ptr(), hn, r_ids.data_ptr(), r_pos.data_ptr()); ts.append(time.perf_counter() - t0)
        print(f"buffers on node {node}, {kind:8s} matchFromHostReduce :", stats(ts))
        if kind == "pinned":
            d_in = torch.empty(hn, dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): d_in.copy_(h_in, non_blocking=True)
            torch.cuda.synchronize()
            print(f"buffers on node {node}: link h2d pinned %.1f GB/s" % (3 * hn / (time.perf_counter() - t0) / 1e9))
            del d_in
        del h_in, h_out, r_ids, r_pos

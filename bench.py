#!/usr/bin/env python3
"""bench.py -- input GB/s scanned by the PFAC match path on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c5|c6] [--size-mib 1024]

One "step" = one PFAC_matchFromDevice() pass over the rank's slice of the synthetic stream, input
and result buffers resident in HBM, transition tables already uploaded.  The stream is sharded as
independent 1 GiB slices with a maxPatternLen+1 read-ahead tail (reference omp_PFAC.cpp:319-377);
there is no data-path collective, RCCL only gathers the per-rank (match count, checksum) facts.

Process layout.  Started plainly (`python bench.py --gpus N`), this process is an ORCHESTRATOR that
never touches the GPU: it builds what is missing, starts N fresh rank processes (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment -- what torch.distributed.run
would set), relays rank 0's JSON line and exits with the worst child's code.  At N = 1 it also runs the
CPU baseline in its own process (own OpenMP environment, GPU idle) and the two rocprofv3 PMC passes
that measure the HBM traffic of the dominant kernel.  Started by `python -m torch.distributed.run`
(RANK is set), the process is a rank.

Rank 0 prints ONE JSON line (the driver contract) that also carries
  "roofline":      algorithmic HBM bytes (5 B per input byte: 1 read + 4 written) / kernel time,
                   kernel time measured with HIP events on the launch stream inside the timed region;
                   "traffic" from the PMC passes of this run (or of the committed profile, see
                   "traffic_source")
  "cpu_baseline":  the reference's own OpenMP matcher (oracle/_ref, compiled from the unmodified
                   reference sources) timed on this host's cores on a bounded sample, N = 1 only
  "other_configs": the other single-GPU BASELINE configurations (c2 texture on/off, c5 dense/hashed), c6 = the c5 stream over
                   c3's set + c5's shared-prefix patterns (the worst input on the FULL set), the reference-layout kernels, c4 as
                   one 8 GiB workload; 10 launches each on the same GPU, N = 1 only
  "call_size_sweep": 1 / 4 / 16 / 64 / 256 MiB calls of the headline stream through PFACX_KERNEL_AUTO, and one cold call after
                   a second of idle GPU, N = 1 only

Bit-exactness (outside the timed region): every rank compares its result with the committed digest of
the REFERENCE's CPU/OMP result for its slice (tests/golden/full_digests.json: match count, position
checksum, FNV-1a-64 of the int32 vector); at sizes without a digest, with the oracle on sampled
windows.  At N = 1 the CPU baseline's result (the reference matcher itself) is also compared with the
GPU result over the whole CPU sample.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

# OpenMP workers of the rank processes (numpy/torch pools, the oracle's window checks before the timed
# region) must sleep, not spin, once their parallel region is over: spinning host threads delay the
# launches of the timed steps.  The CPU-baseline worker replaces these (cpu_worker_env).
if "cpu" not in sys.argv[1:]:          # not the --worker cpu process: it runs with the OpenMP runtime's own wait policy
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
ALGO_BYTES_PER_INPUT_BYTE = 5    # 1 B input read + 4 B int32 result written
SETTLE_STEPS = 32                # untimed launches after host-side checks, before the warmup steps
OTHER_STEPS = 10                 # timed launches per entry of "other_configs"
PMC_LAUNCHES = 4                 # launches of --worker pmc that the counter means are taken over (behind one synchronised launch)
HOST_CALLS = 20                  # timed PFAC_matchFromHost / ...Reduce calls per buffer kind (median, best, p90)
DIGESTS = os.path.join(ROOT, "tests", "golden", "full_digests.json")
SCRATCH = os.path.join(ROOT, "gpurun_out", "bench")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c5", "c6"])
    ap.add_argument("--size-mib", type=int, default=1024, help="input bytes per GPU, MiB")
    ap.add_argument("--variant", default="auto", choices=["auto", "filter", "naive", "reftable"],
                    help="auto = the library default (PFACX_KERNEL_AUTO: what a drop-in PFAC.h caller gets; the filter kernel at bench sizes)")
    ap.add_argument("--walker", default="auto", choices=["auto", "window", "stage", "veto"],
                    help="walker of the full-result filter kernel (PFACX_setWalker): auto = the library default, what the handle's previous launch found its stream to be")
    ap.add_argument("--texture", default="auto", choices=["auto", "on", "off"])
    ap.add_argument("--perf-mode", default=None, choices=[None, "dense", "hash"])
    ap.add_argument("--platform", default="gpu", choices=["gpu", "cpu_omp"],
                    help="cpu_omp: PFAC_setPlatform(PFAC_PLATFORM_CPU_OMP) + PFAC_matchFromHost -- a dry run of the rank/"
                         "shard/gather logic on a machine without a GPU (never the reported metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--spread", action="store_true", help="measure the placement spread (4 buffer pairs) even with --no-other-configs")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"],
                    help="auto: measure HBM traffic of the scan kernel with two rocprofv3 --pmc passes (N = 1, orchestrated runs)")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="target duration of ONE CPU-baseline run (3 runs are timed)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) is the real path; gloo lets several ranks share one GPU for a dry run")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the facts all-gather even with one rank (one RCCL all_gather on "
                         "hardware without a multi-GPU node: tests/test_multi_gpu.py)")
    ap.add_argument("--worker", default=None, choices=[None, "cpu", "pmc", "rank"],
                    help="rank: be the single rank right here, no orchestrator and no child process (the form to put behind `rocprofv3 ... --`); "
                         "pmc: the same, a few launches only; cpu: the CPU-baseline process")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, what the driver's --gpus N runs): one slice of --size-mib per GPU.  strong: BASELINE config 4 as ONE workload -- "
                         "--total-gib GiB of the stream in slices of --size-mib, dealt round-robin over the ranks (rank r scans slices r, r + N, ...: "
                         "PFAC/test/omp_PFAC.cpp:351-355), the same total work at N = 1, 2, 4, 8")
    ap.add_argument("--total-gib", type=int, default=8, help="--scaling strong: size of the whole stream")
    ap.add_argument("--total-mib", type=int, default=None, help="... in MiB (small dry runs: tests/test_sharding_gloo.py)")
    ap.add_argument("--sparse-file", default=None, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------- digests

def load_digest(workload, slice_index, size_mib, last):
    """Committed digest of the reference's CPU/OMP result for one slice (tests/golden/make_full_digests.py)."""
    try:
        doc = json.load(open(DIGESTS))
    except OSError:
        return None
    for r in doc["records"]:
        if r["workload"] == workload and r["slice"] == slice_index and r["size_mib"] == size_mib:
            d = dict(r["last" if last else "inner"])
            d["input_fnv1a"] = r["input_fnv1a"]
            d["variant"] = "last" if last else "inner"
            return d
    return None


def sparse_result(d_out, n):
    """(positions, ids) of the non-zero results of a device int32 result vector."""
    import torch
    pos = torch.nonzero(d_out[:n]).flatten()
    ids = d_out[:n][pos]
    return pos.cpu().numpy().astype(np.int64), ids.cpu().numpy().astype(np.int32)


def check_against_digest(dg, pos, ids, n, base, host_in):
    """The result vector is (pos, ids) and zero elsewhere: compare with the reference digest."""
    from pfac_amd import sharding
    from pfac_amd import workloads as wl
    facts = {
        "match_count": int(pos.size),
        "checksum": int(sharding.position_checksum(pos, ids, base=base)),
        "fnv1a64": int(wl.fnv1a_sparse_i32(pos, ids, n)),
    }
    ok = all(facts[k] == dg[k] for k in facts)
    same_input = wl.fnv1a(host_in[:n]) == dg["input_fnv1a"]
    if not same_input:
        log("[verify] the generated input differs from the one the digest was made from (generator drift)")
    return ok and same_input, facts


def check_against_oracle_windows(ob, pattern_file, host_in, d_out, n, n_read, perf_mode):
    """No digest for this size: the oracle re-scans sampled 1 MiB windows (+ overlap)."""
    oracle = ob.Oracle(pattern_file, dense=(perf_mode == 0), hashed=(perf_mode == 1))
    rng = np.random.Generator(np.random.PCG64(99))
    win = 1 << 20
    tail = oracle.max_pattern_len + 1
    starts = [0, max(0, n - win)] + [int(x) for x in rng.integers(0, max(1, n - win), size=6)]
    ok = True
    for s in starts:
        e = min(n, s + win)
        r = min(n_read, e + tail)
        want = oracle.match(host_in[s:r], hashed=(perf_mode == 1), omp=True)[: e - s]
        got = d_out[s:e].cpu().numpy() if hasattr(d_out, "cpu") else d_out[s:e]
        if not np.array_equal(got, want):
            ok = False
            log(f"[verify] MISMATCH in window [{s},{e})")
    oracle.close()
    return ok


# ----------------------------------------------------------------------------- PMC traffic

def committed_traffic(workload, kernel_substr):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_hbm_traffic_<workload>.json):
    2 x FETCH_SIZE (gfx950 reports half of coalesced reads, MI355X_MICROARCH.md "HBM") + WRITE_SIZE."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_hbm_traffic_{workload}.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))

        def mine(k):   # pfac_scan_filter<MODE, HAS_SHORT, REDUCE, ...>: the full-result kernel has REDUCE = false
            targs = [a.strip() for a in k.split("<", 1)[-1].split(">", 1)[0].split(",")]
            return kernel_substr in k and (len(targs) < 3 or targs[2] == "false")
        f = [v for k, v in d["FETCH_SIZE"]["scan"].items() if mine(k)]
        w = [v for k, v in d["WRITE_SIZE"]["scan"].items() if mine(k)]
        if not f or not w:
            return None, None
        return int((2.0 * f[0] + w[0]) * 1024), os.path.relpath(files[-1], ROOT) + " (committed profile, not this run)"
    except Exception:
        return None, None


def pmc_counter_mean(csv_dir, kernel_substr):
    """Mean Counter_Value over the dispatches of the full-result scan kernel in a rocprofv3 counter CSV."""
    import csv
    import glob
    vals = []
    for path in glob.glob(os.path.join(csv_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if kernel_substr in r["Kernel_Name"]:
                vals.append((int(r.get("Dispatch_Id") or len(vals)), float(r["Counter_Value"])))
    vals = [v for _, v in sorted(vals)][-PMC_LAUNCHES:]       # the launches behind the worker's first, synchronised one
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def measure_traffic(argv_common, kernel_substr):
    """Two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) over a short run of the
    same workload; the profiled program is this script in --worker pmc mode, placed directly after `--`."""
    import shutil
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    out = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(SCRATCH, "pmc_" + counter)
        shutil.rmtree(d, ignore_errors=True)
        cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--worker", "pmc"] + argv_common
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {counter} timed out"
        mean, count = pmc_counter_mean(d, kernel_substr)
        if p.returncode != 0 or mean is None:
            return None, f"rocprofv3 --pmc {counter} rc {p.returncode}, {count} dispatches"
        out[counter] = (mean, count)
    kb_fetch, kb_write = out["FETCH_SIZE"][0], out["WRITE_SIZE"][0]
    return {"bytes": int((2.0 * kb_fetch + kb_write) * 1024), "fetch_size_kb": round(kb_fetch, 1), "write_size_kb": round(kb_write, 1),
            "dispatches": out["FETCH_SIZE"][1]}, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (2 x FETCH_SIZE + WRITE_SIZE, KB = 1024 B)"


# ------------------------------------------------------------------------------ CPU baseline

def lscpu_facts():
    facts = {}
    try:
        for line in subprocess.run(["lscpu"], stdout=subprocess.PIPE, text=True, timeout=10).stdout.splitlines():
            k, _, v = line.partition(":")
            facts[k.strip()] = v.strip()
    except Exception:
        pass

    def num(k, default):
        try:
            return int(facts.get(k, default))
        except ValueError:
            return default
    sockets, cps, tpc = num("Socket(s)", 1), num("Core(s) per socket", 0), num("Thread(s) per core", 1)
    logical = os.cpu_count() or 1
    physical = sockets * cps if cps else max(1, logical // max(1, tpc))
    try:
        physical = min(physical, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return {"model": facts.get("Model name", "unknown"), "sockets": sockets, "physical_cores": max(1, physical),
            "logical_cpus": logical, "threads_per_core": tpc}


def cpu_worker_env(threads):
    """OpenMP environment of the CPU-baseline process (SURVEY 8d / BASELINE.md 3): one thread per
    physical core, bound close to cores; the default (active) wait policy of the OpenMP runtime."""
    env = dict(os.environ)
    for k in ("OMP_WAIT_POLICY", "GOMP_SPINCOUNT"):
        env.pop(k, None)
    env.update(OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores")
    return env


def cpu_worker(args):
    """--worker cpu: time the reference's OpenMP matcher (oracle/_ref; the C port if _ref is absent) on a
    bounded prefix of the rank-0 stream, 3 runs; compare its result with the GPU's sparse result."""
    facts = lscpu_facts()              # before the OpenMP runtime binds this thread to one core (OMP_PROC_BIND)
    from oracle import binding as ob   # cpu_baseline leg
    from pfac_amd import workloads as wl
    cfg = wl.make_config(args.workload)
    perf_mode = cfg.perf_mode if args.perf_mode is None else (0 if args.perf_mode == "dense" else 1)
    os.makedirs(SCRATCH, exist_ok=True)
    pf = wl.write_pattern_file(os.path.join(SCRATCH, f"{cfg.name}_cpu.pat"), cfg.patterns)
    oracle = ob.Oracle(pf, dense=(perf_mode == 0), hashed=(perf_mode == 1))
    use_ref = ob.have_reference()
    if perf_mode == 0:
        dense = oracle.dense_table()
    else:
        row, val = oracle.hash_row(), oracle.hash_val()
    n = args.size_mib << 20
    host_in = cfg.input_slice(n, 0)

    def run(sample):
        t0 = time.perf_counter()
        if use_ref:
            if perf_mode == 0:
                r = ob.Reference.match_dense(sample, dense, oracle.num_patterns, oracle.initial_state, omp=True)
            else:
                r = ob.Reference.match_hash(sample, row, val, oracle.num_patterns, oracle.initial_state, omp=True)
        else:
            r = oracle.match(sample, hashed=(perf_mode == 1), omp=True)
        return time.perf_counter() - t0, r

    pilot = min(n, 16 << 20)
    t, _ = run(host_in[:pilot])
    sample = int(min(n, max(pilot, pilot / t * args.cpu_seconds)))
    if sample > (1 << 20):
        sample -= sample % (1 << 20)
    times = []
    for _ in range(3):
        t, result = run(host_in[:sample])
        times.append(t)
    fn = ("PFAC_CPU_OMP_spaceDriven" if perf_mode else "PFAC_CPU_OMP_timeDriven") if use_ref else "oracle C port"
    threads = ob.omp_max_threads()
    out = {
        "value": round(sample / min(times) / 1e9, 4), "unit": "GB/s", "cores": threads,
        "kind": "reference" if use_ref else "port",
        "best": round(sample / min(times) / 1e9, 4), "median": round(sample / float(np.median(times)) / 1e9, 4),
        "runs_s": [round(x, 4) for x in times],
        "sample": f"first {sample >> 20} MiB of the rank-0 {cfg.name} stream, {fn}"
                  f" ({'reference sources compiled unmodified, oracle/_ref' if use_ref else 'oracle/pfac_oracle.c'}), {threads} OpenMP threads, "
                  f"3 runs after a 16 MiB pilot, value = best",
        "cpu": facts,
        "omp_env": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES", "OMP_WAIT_POLICY")},
    }
    # cross-check: the reference matcher's result == the GPU's result, over every position of the sample
    # whose walk cannot reach the end of the sample (the GPU scanned the whole stream)
    if args.sparse_file and os.path.exists(args.sparse_file):
        z = np.load(args.sparse_file)
        limit = sample if sample == n and int(z["n_read"]) == n else sample - oracle.max_pattern_len
        cpu_pos = np.flatnonzero(result[:limit])
        keep = z["pos"] < limit
        same = bool(np.array_equal(cpu_pos, z["pos"][keep]) and np.array_equal(result[cpu_pos], z["ids"][keep]))
        out["gpu_result_equals_cpu_result"] = same
        out["compared_positions"] = int(limit)
        out["compared_matches"] = int(cpu_pos.size)
    oracle.close()
    print(json.dumps(out), flush=True)
    return 0


# ---------------------------------------------------------------------------------- rank process

class Run:
    """One workload on one device: handle, input slice, device buffers."""

    def __init__(self, args, name, perf_mode, texture, rank, world, device, buffers=None, share=None, host_in=None):
        """rank / world: slice `rank` of a stream of `world` slices.  share = another Run: its handle (same pattern set) and
        its result buffer are used, only the input is this Run's own (the slices of --scaling strong)."""
        import torch
        from pfac_amd import api, sharding
        from pfac_amd import workloads as wl
        self.api, self.torch = api, torch
        self.gpu = args.platform == "gpu"
        self.owner = share is None
        if share is None:
            self.cfg = wl.make_config(name)
            self.perf_mode = self.cfg.perf_mode if perf_mode is None else perf_mode
            os.makedirs(SCRATCH, exist_ok=True)
            self.pattern_file = wl.write_pattern_file(os.path.join(SCRATCH, f"{self.cfg.name}_rank{rank}.pat"), self.cfg.patterns)
            if self.gpu:
                self.handle = api.PFAC.create()
            else:
                self.handle = api.PFAC.createHostOnly()
                self.handle.setPlatform(api.PFAC_PLATFORM_CPU_OMP)
            self.handle.setPerfMode(self.perf_mode)
            self.handle.setTextureMode({"auto": api.PFAC_AUTOMATIC, "on": api.PFAC_TEXTURE_ON, "off": api.PFAC_TEXTURE_OFF}[texture])
            self.variant = {"auto": api.PFACX_KERNEL_AUTO, "filter": api.PFACX_KERNEL_FILTER, "naive": api.PFACX_KERNEL_NAIVE,
                            "reftable": api.PFACX_KERNEL_REFTABLE}[args.variant]
            if args.variant != "auto":                             # auto: the handle keeps the library default, nothing is set
                self.handle.setKernelVariant(self.variant)
            if getattr(args, "walker", "auto") != "auto" and self.gpu:
                self.handle.setWalker({"window": api.PFACX_WALKER_WINDOW, "stage": api.PFACX_WALKER_STAGE, "veto": api.PFACX_WALKER_VETO}[args.walker])
            self.handle.readPatternFromFile(self.pattern_file)
            self.info = self.handle.info()
        else:
            self.cfg, self.perf_mode, self.pattern_file, self.handle, self.info = share.cfg, share.perf_mode, share.pattern_file, share.handle, share.info
        # slice `rank` of the N x size stream plus the head of the next slice (generators are prefix-stable)
        if host_in is None:
            self.host_in, self.n = sharding.rank_input(self.cfg, args.size_mib << 20, rank, world, self.info.maxPatternLen)
        else:
            self.host_in, self.n = host_in, args.size_mib << 20
        self.n_read = self.host_in.size
        if share is not None:
            if self.gpu:
                self.d_in = torch.from_numpy(self.host_in).to(device)
                self.d_out = share.d_out
                assert self.d_out.numel() >= self.n_read
                torch.cuda.synchronize()
            else:
                self.h_out = share.h_out
        elif self.gpu:
            if buffers is not None and buffers[0].numel() >= self.n_read:
                self.d_in, self.d_out = buffers
                self.d_in[: self.n_read].copy_(torch.from_numpy(self.host_in))
                self.d_out.fill_(-1)
            else:
                self.d_in = torch.from_numpy(self.host_in).to(device)
                self.d_out = torch.full((self.n_read,), -1, dtype=torch.int32, device=device)
            torch.cuda.synchronize()
        else:
            self.h_out = np.full(self.n_read, -1, dtype=np.int32)

    def step(self):
        if self.gpu:
            self.handle.matchFromDevice(self.d_in.data_ptr(), self.n_read, self.d_out.data_ptr())
        else:
            self.handle.matchFromHost(self.host_in.ctypes.data, self.n_read, self.h_out.ctypes.data)

    def sparse(self):
        if self.gpu:
            self.torch.cuda.synchronize()
            return sparse_result(self.d_out, self.n)
        pos = np.flatnonzero(self.h_out[: self.n])
        return pos.astype(np.int64), self.h_out[pos]

    def verify(self, args, rank, world):
        """-> (bit_exact, method, pos, ids)"""
        pos, ids = self.sparse()
        if args.no_verify:
            return True, "not verified (--no-verify)", pos, ids
        dg = load_digest(self.cfg.name, rank, args.size_mib, last=(rank == world - 1))
        if dg is not None:
            ok, facts = check_against_digest(dg, pos, ids, self.n, rank * self.n, self.host_in)
            if not ok:
                log(f"[verify r{rank}] result {facts} != reference digest { {k: dg[k] for k in facts} }")
            return ok, "== digest of the reference's PFAC_CPU_OMP result (tests/golden/full_digests.json: count, checksum, FNV-1a-64)", pos, ids
        from oracle import binding as ob   # checker only
        out = self.d_out if self.gpu else self.h_out
        ok = check_against_oracle_windows(ob, self.pattern_file, self.host_in, out, self.n, self.n_read, self.perf_mode)
        return ok, "== oracle on 8 sampled 1 MiB windows (no committed digest for this size)", pos, ids

    def timed(self, steps, warmup, settle, barrier=None):
        """-> (kernel_ms per step from HIP events on the launch stream, wall seconds of the K steps)"""
        for _ in range(settle + warmup):
            self.step()
        if not self.gpu:
            if barrier:
                barrier()
            ms = []
            t0 = time.perf_counter()
            for _ in range(steps):
                t1 = time.perf_counter()
                self.step()
                ms.append((time.perf_counter() - t1) * 1e3)
            if barrier:
                barrier()
            return ms, time.perf_counter() - t0
        from pfac_amd import hiprt
        torch = self.torch
        events = [(hiprt.Event(), hiprt.Event()) for _ in range(steps)]
        torch.cuda.synchronize()
        if barrier:
            barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b in events:
            a.record(0)
            self.step()
            b.record(0)
        torch.cuda.synchronize()
        if barrier:
            barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        return [a.elapsed_ms(b) for a, b in events], elapsed

    def scan_stats(self):
        """Counters of the last launch (pfac_ext.h: PFACX_getScanStats), or None."""
        if not self.gpu:
            return None
        st = self.handle.scanStats(self.n_read)
        return st if st["walkerRounds"] else None

    def close(self):
        if self.owner:
            self.handle.destroy()


class SliceSet:
    """BASELINE config 4 as ONE workload (--scaling strong): the stream is `total` slices of --size-mib (slice i from seed + i,
    scanned with the maxPatternLen + 1 bytes of its successor: omp_PFAC.cpp:324), of which this rank scans slices rank,
    rank + world, ... (omp_PFAC.cpp:351-355: tid + k * num_threads; pfac_amd/sharding.py: rank_slices).  One handle per rank,
    every slice's input resident in HBM, one result buffer (a slice's results are checked before the next launch overwrites
    them: the correctness gate runs slice by slice)."""

    def __init__(self, args, name, perf_mode, texture, total, rank, world, device):
        import concurrent.futures
        from pfac_amd import sharding
        self.total, self.rank, self.world = total, rank, world
        self.mine = sharding.rank_slices(total, rank, world)
        assert self.mine, "more ranks than slices"
        n = args.size_mib << 20
        # the rank's handle and result buffer come with its first slice
        owner = Run(args, name, perf_mode, texture, self.mine[0], total, device)
        room = n + sharding.overlap_bytes(owner.info.maxPatternLen)
        if owner.gpu and owner.d_out.numel() < room:
            owner.d_out = owner.torch.full((room,), -1, dtype=owner.torch.int32, device=device)
        elif not owner.gpu and owner.h_out.size < room:
            owner.h_out = np.full(room, -1, dtype=np.int32)
        self.owner = owner
        self.runs, self.index = [owner], [self.mine[0]]
        rest = self.mine[1:]
        with concurrent.futures.ThreadPoolExecutor(4) as pool:       # the generator is C code: slices are generated side by side
            hosts = list(pool.map(lambda i: sharding.rank_input(owner.cfg, n, i, total, owner.info.maxPatternLen)[0], rest))
        for i, h in zip(rest, hosts):
            self.runs.append(Run(args, name, perf_mode, texture, i, total, device, share=owner, host_in=h))
            self.index.append(i)

    def gate(self, args):
        """every slice once, each checked against its reference digest: -> (all exact, method, matches, folded checksum)"""
        from pfac_amd import sharding
        ok_all, method, count, acc = True, None, 0, 0
        for r, i in zip(self.runs, self.index):
            r.step()
            ok, method, pos, ids = r.verify(args, i, self.total)
            ok_all = ok_all and ok
            count += int(pos.size)
            acc = (acc + sharding.position_checksum(pos, ids, base=i * r.n)) & 0xFFFFFFFFFFFFFFFF
        return ok_all, method, count, acc

    def step(self):
        for r in self.runs:
            r.step()

    def timed(self, steps, warmup, settle, barrier=None):
        """-> (ms per step = one pass over this rank's slices, wall seconds of the K steps)"""
        gpu = bool(self.runs) and self.runs[0].gpu
        for _ in range(settle + warmup):
            self.step()
        if gpu:
            from pfac_amd import hiprt
            torch = self.runs[0].torch
            events = [(hiprt.Event(), hiprt.Event()) for _ in range(steps)]
            torch.cuda.synchronize()
        if barrier:
            barrier()
        t0 = time.perf_counter()
        ms = []
        for k in range(steps):
            if gpu:
                events[k][0].record(0)
                self.step()
                events[k][1].record(0)
            else:
                t1 = time.perf_counter()
                self.step()
                ms.append((time.perf_counter() - t1) * 1e3)
        if gpu:
            torch.cuda.synchronize()
        if barrier:
            barrier()
        elapsed = time.perf_counter() - t0
        if gpu:
            ms = [a.elapsed_ms(b) for a, b in events]
        return ms, elapsed

    def close(self):
        for r in self.runs:
            r.close()


def kernel_name(args, run):
    """The kernel a launch of this run goes to (the library default takes the filter kernel from 32 MiB up)."""
    if args.variant == "reftable":
        return "pfac_scan_tiled<REF>"          # the tiled frame over the reference-layout table (scan_tiled.hip: launchTiledRef)
    if args.variant == "naive" or (args.variant == "auto" and run.n_read < (32 << 20)):
        return "pfac_scan_tiled"
    return "pfac_scan_filter"


def walker_table(args, run):
    """What the kernel walks: the filter kernel and the tiled kernel walk the device-only chained table in BOTH perf modes;
    only the reference-shaped kernel (--variant reftable) walks the reference-layout table that `table` names."""
    return "chained (device-only, both perf modes)" if args.variant != "reftable" else ("hashed rowPtr/valPtr (reference layout)" if run.perf_mode else "dense int[S][256] (reference layout)")


_BUILD = None


def build_info():
    """Compile-time shape of the kernel module (PFACX_buildInfo of libpfac_gfx950.so)."""
    global _BUILD
    if _BUILD is None:
        import ctypes as C
        from pfac_amd import api
        try:
            mod = C.CDLL(api.library_paths()[1])
            mod.PFACX_buildInfo.restype = C.c_char_p
            _BUILD = mod.PFACX_buildInfo().decode()
        except (OSError, AttributeError):
            _BUILD = "unknown"
    return _BUILD


def roofline_block(kernel_ms, n_read, kname):
    avg_s = float(np.mean(kernel_ms)) / 1e3
    achieved = ALGO_BYTES_PER_INPUT_BYTE * n_read / avg_s / 1e9
    return {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
        "kernel": kname,
        "kernel_ms_avg": round(avg_s * 1e3, 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
        "kernel_ms_median": round(float(np.median(kernel_ms)), 4), "kernel_ms_max": round(float(np.max(kernel_ms)), 4),
        "kernel_ms_steps": [round(float(x), 3) for x in kernel_ms] if len(kernel_ms) <= 64 else None,
        "algorithmic_bytes_per_launch": ALGO_BYTES_PER_INPUT_BYTE * n_read,
        "input_only_frac": round(n_read / avg_s / 1e9 / HBM_PEAK_GBS, 4),
    }


def size_sweep(run):
    """Calls of the sizes an IDS submits (PFAC_algorithm.pdf Table 2 measures 2 MB ... 192 MB): the first 1, 4, 16, 64 and 256 MiB of the
    headline stream through the handle as it stands (PFACX_KERNEL_AUTO: the tiled kernel below 32 MiB, the filter kernel from there on),
    back-to-back calls between two HIP events; and ONE cold call of the whole stream after a second of idle GPU (clocks down: what a
    caller's first call after a pause pays; the headline is a steady-clock figure)."""
    from pfac_amd import hiprt
    torch = run.torch
    out = {}
    for mib in (1, 4, 16, 64, 256):
        n = mib << 20
        if n > run.n_read:
            break
        calls = max(10, min(200, (2048 << 20) // n))
        for _ in range(5):
            run.handle.matchFromDevice(run.d_in.data_ptr(), n, run.d_out.data_ptr())
        torch.cuda.synchronize()
        a, b = hiprt.Event(), hiprt.Event()
        a.record(0)
        for _ in range(calls):
            run.handle.matchFromDevice(run.d_in.data_ptr(), n, run.d_out.data_ptr())
        b.record(0)
        torch.cuda.synchronize()
        ms = a.elapsed_ms(b) / calls
        out[f"{mib}_MiB"] = {"ms_per_call": round(ms, 5), "calls": calls, "input_GBps": round(n / (ms / 1e3) / 1e9, 1),
                             "frac": round(ALGO_BYTES_PER_INPUT_BYTE * n / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                             "kernel_launched": "pfac_scan_tiled" if n < (32 << 20) else "pfac_scan_filter"}
    # the cold call: everything above has been synchronised; a second of idle, then one bracketed call of the whole stream
    for _ in range(3):
        run.step()
    torch.cuda.synchronize()
    warm_a, warm_b = hiprt.Event(), hiprt.Event()
    warm_a.record(0)
    run.step()
    warm_b.record(0)
    torch.cuda.synchronize()
    colds = []
    for _ in range(3):
        time.sleep(1.0)
        a, b = hiprt.Event(), hiprt.Event()
        a.record(0)
        run.step()
        b.record(0)
        torch.cuda.synchronize()
        colds.append(a.elapsed_ms(b))
    out["cold_call"] = {"bytes": int(run.n_read), "ms_after_1s_idle": [round(x, 4) for x in colds], "ms_warm": round(warm_a.elapsed_ms(warm_b), 4),
                        "cold_over_warm": round(float(np.median(colds)) / warm_a.elapsed_ms(warm_b), 3),
                        "note": "one call of the whole stream bracketed by HIP events after 1 s of idle GPU (three times), against the same call behind three warm ones"}
    return out


def other_configs(args, device, buffers):
    """The other single-GPU BASELINE configurations on the same GPU and the same buffers:
    c2 with texture mode on and off, c5 with the dense and the hashed table."""
    from pfac_amd import api
    out = {}
    todo = [("c2_texture_on", "c2", None, "on", None), ("c2_texture_off", "c2", None, "off", None),
            ("c5_dense", "c5", api.PFAC_TIME_DRIVEN, "auto", None), ("c5_hashed", "c5", api.PFAC_SPACE_DRIVEN, "auto", None),
            # the worst input on the FULL set (PFAC_hash_draft.pdf Table 5 measures its "worst" input on the same Snort set as its regular one): C5's near-miss
            # stream over C3's 30 000 patterns + C5's 1 000 shared-prefix patterns.  The handle finds the veto kernel by itself (PFACX_WALKER_AUTO: settling launches)
            ("c6_near_miss_stream_over_the_snort_scale_set", "c6", api.PFAC_SPACE_DRIVEN, "auto", None),
            # BASELINE config 2 names the dense 2-D table: the filter kernel walks the chained table in both perf modes, so this is the
            # entry in which a kernel really walks int[S][256] -- the reference-shaped one (PFACX_KERNEL_REFTABLE), and the tiled kernel alone
            ("c2_dense_table_reference_shaped_kernel", "c2", api.PFAC_TIME_DRIVEN, "on", "reftable"),
            ("c2_dense_table_reference_shaped_kernel_texture_off", "c2", api.PFAC_TIME_DRIVEN, "off", "reftable"),
            # ... and BASELINE config 3's hashed rowPtr / valPtr pair, walked with two dependent loads per byte (PFAC_kernel_spaceDriven.cu:76-124)
            ("c3_hashed_table_reference_shaped_kernel", "c3", api.PFAC_SPACE_DRIVEN, "on", "reftable"),
            ("c2_tiled_kernel", "c2", None, "auto", "naive")]
    for key, name, perf, tex, variant in todo:
        t0 = time.perf_counter()
        if variant is not None:
            import copy
            args = copy.copy(args)
            args.variant = variant
        run = Run(args, name, perf, tex, 0, 1, device, buffers)
        run.step()
        ok, method, pos, _ = run.verify(args, 0, 1)
        ms, _ = run.timed(OTHER_STEPS, 2, SETTLE_STEPS)             # the check has left the GPU idle for seconds: the same settling as the headline (a kernel bound by
                                                                       # instruction issue, like C5's, is still slow 20 launches later)
        r = roofline_block(ms, run.n_read, kernel_name(args, run))
        entry = {"workload": run.cfg.description.replace("dense table", "hashed table") if run.perf_mode else run.cfg.description,
                 "table": "hashed" if run.perf_mode else "dense", "walker_table": walker_table(args, run),
                 "texture_mode": int(run.handle.info().textureMode), "steps": OTHER_STEPS, "kernel": args.variant, "kernel_launched": kernel_name(args, run),
                 "kernel_ms_avg": r["kernel_ms_avg"], "kernel_ms_min": r["kernel_ms_min"], "frac": r["frac"],
                 "input_GBps": round(run.n_read / (r["kernel_ms_avg"] / 1e3) / 1e9, 1),
                 "matches": int(pos.size), "bit_exact": bool(ok), "bit_exact_method": method}
        st = run.scan_stats()
        if st:
            entry["walk_stats"] = st
        if name == "c6":
            entry["note"] = ("the VETO = 2 kernel runs in one of two classes by where the driver placed the process's buffers: ~1.07 ms (0.63) or ~1.16 ms (0.58); "
                             "roofline.bare_stream_1r4w.ms of this line -- the same buffers -- shows which (~0.90 / ~0.94): profiles/r06_c6_placement_classes.txt")
        out[key] = entry
        run.close()
        log(f"[bench] {key}: {entry['kernel_ms_avg']} ms, frac {entry['frac']}, exact {ok} ({time.perf_counter() - t0:.1f}s)")
    return out


def placement_facts(args, device):
    """(PCI bus id as an integer, NUMA node of the device or -- CPU dry run -- of the CPU this rank runs on): two of the int64 facts every rank
    contributes, so that the first run on a real 8-GPU node records which device and which socket each rank had (omp_PFAC.cpp:257-290 binds
    one host thread per device the same way and says nothing about where)."""
    if args.platform == "gpu" and device is not None:
        from pfac_amd import hiprt
        bus = hiprt.pci_bus_id(int(str(device).split(":")[-1]))
        return hiprt.pci_code(bus), hiprt.numa_node_of_pci(bus)
    node = -1
    try:
        cpu = sorted(os.sched_getaffinity(0))[0]
        for d in os.listdir("/sys/devices/system/node"):
            if d.startswith("node") and os.path.exists(f"/sys/devices/system/node/{d}/cpu{cpu}"):
                node = int(d[4:])
    except (OSError, ValueError, AttributeError, IndexError):
        pass
    return -1, node


def rank_placement(allf):
    """the gathered placement facts as the JSON line reports them"""
    from pfac_amd import hiprt
    return [{"rank": int(r[4]), "pci_bus_id": hiprt.pci_string(int(r[5])) or None, "numa_node": int(r[6])} for r in allf]


def setup_rank(args):
    """Rank, device and process group of this process: -> None (refused) or (rank, world, device, use_dist, barrier, dist)"""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a number for a different rank count")
        return None
    gpu = args.platform == "gpu"

    import torch
    import torch.distributed as dist
    device = None
    if gpu:
        if not torch.cuda.is_available():
            log("[bench] no GPU: the match path has no CPU fallback (use --platform cpu_omp --dist-backend gloo for a dry run)")
            return None
        ndev = torch.cuda.device_count()
        if args.dist_backend == "gloo":
            local_rank = local_rank % ndev                       # dry run: ranks may share a device
        elif local_rank >= ndev:
            log(f"[bench] rank {rank} needs GPU {local_rank} but only {ndev} visible: --gpus {args.gpus} needs {args.gpus} GPUs")
            return None
        torch.cuda.set_device(local_rank)
        device = f"cuda:{local_rank}"
    elif args.dist_backend != "gloo":
        log("[bench] --platform cpu_omp needs --dist-backend gloo")
        return None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if args.force_dist and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    def barrier():
        if use_dist:
            dist.barrier()

    return rank, world, device, use_dist, barrier, dist


def strong_numbers(args, sset, ms, elapsed_max, total_matches, folded, all_ok, method, world):
    """The facts of one --scaling strong measurement (also other_configs.c4_8gib_one_gpu of the default line)."""
    from pfac_amd import sharding
    n = args.size_mib << 20
    expected = None
    dgs = [load_digest(sset.owner.cfg.name, i, args.size_mib, last=(i == sset.total - 1)) for i in range(sset.total)]
    if all(d is not None for d in dgs):
        ec, es = sharding.combine_checksums([(d["match_count"], d["checksum"]) for d in dgs])
        expected = {"match_count": ec, "checksum": es & 0x7FFFFFFFFFFFFFFF,
                    "equal": bool(ec == total_matches and (es & 0x7FFFFFFFFFFFFFFF) == (folded & 0x7FFFFFFFFFFFFFFF))}
        all_ok = all_ok and expected["equal"]
    steps = len(ms)
    per_launch = [x / len(sset.runs) for x in ms]
    return {"total_bytes": sset.total * n, "slices": sset.total, "slice_bytes": n, "slices_of_rank0": sset.index, "steps": steps,
            "aggregate_GBps": round(sset.total * n / (elapsed_max / steps) / 1e9, 2), "ms_per_step": round(elapsed_max / steps * 1e3, 4),
            "kernel_ms_per_slice_avg": round(float(np.mean(per_launch)), 4),
            "frac": round(ALGO_BYTES_PER_INPUT_BYTE * sset.runs[0].n_read / (float(np.mean(per_launch)) / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
            "matches": int(total_matches), "bit_exact": bool(all_ok), "bit_exact_method": method,
            "folded_result": {"match_count": int(total_matches), "checksum": folded & 0x7FFFFFFFFFFFFFFF}, "folded_reference": expected,
            "note": f"BASELINE config 4 as one workload: {sset.total} slices, each scanned with the head of its successor, dealt round-robin over "
                    f"{world} rank(s) (omp_PFAC.cpp:351-355); inputs resident in HBM, one result buffer per rank"}, all_ok


def strong_rank_main(args):
    """--scaling strong: the same --total-gib stream at every N (see SliceSet)."""
    got = setup_rank(args)
    if got is None:
        return 2
    rank, world, device, use_dist, barrier, dist = got
    gpu = args.platform == "gpu"
    from pfac_amd import sharding
    total_mib = args.total_mib if args.total_mib else args.total_gib << 10
    total = total_mib // args.size_mib
    if total < world or total < 1:
        log(f"[bench] --scaling strong: {total} slice(s) of {args.size_mib} MiB for {world} rank(s)")
        return 2
    t_setup = time.perf_counter()
    perf_mode = None if args.perf_mode is None else (0 if args.perf_mode == "dense" else 1)
    sset = SliceSet(args, args.workload, perf_mode, args.texture, total, rank, world, device)
    log(f"[bench r{rank}] strong: slices {sset.index} of {total}, setup {time.perf_counter() - t_setup:.1f}s")
    ok, method, count, checksum = sset.gate(args)
    ms, elapsed = sset.timed(args.steps, args.warmup, SETTLE_STEPS if gpu else 0, barrier)
    if not args.no_verify:                                     # the result of the last launch again (the rank's last slice)
        last, i = sset.runs[-1], sset.index[-1]
        ok2, _, _, _ = last.verify(args, i, total)
        ok = ok and ok2
    allf = sharding.all_gather_facts([count, checksum & 0x7FFFFFFFFFFFFFFF, int(ok), int(elapsed * 1e9), rank, *placement_facts(args, device)],
                                     device=device if (gpu and args.dist_backend == "nccl") else None, force=args.force_dist)
    elapsed_max = float(allf[:, 3].max()) / 1e9
    total_matches, folded = sharding.combine_checksums([(int(c), int(s)) for c, s in allf[:, :2]])
    all_ok = bool(allf[:, 2].all()) and sorted(int(r) for r in allf[:, 4]) == list(range(world))
    rc = 0
    if rank == 0:
        facts, all_ok = strong_numbers(args, sset, ms, elapsed_max, total_matches, folded, all_ok, method, world)
        info, run0 = sset.owner.info, sset.owner
        kname = kernel_name(args, run0)
        per_launch = [x / len(sset.runs) for x in ms]
        out = {
            "metric": "input GB/s scanned (PFAC_matchFromDevice, bit-exact)" if gpu else
                      "input GB/s scanned (DRY RUN on the CPU_OMP platform, not the metric)",
            "value": facts["aggregate_GBps"], "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "settle_steps": SETTLE_STEPS if gpu else 0, "ms_per_step": facts["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": f"c4 = {run0.cfg.name}: {run0.cfg.description}; {total_mib} MiB in {total} slices of {args.size_mib} MiB "
                            f"(overlap {sharding.overlap_bytes(info.maxPatternLen)} B) dealt round-robin over {world} GPU(s): a step is one pass over the whole stream",
                "patterns": info.numOfPatterns, "states": info.numOfStates, "table": "hashed" if run0.perf_mode else "dense",
                "table_bytes": int(info.sizeOfTableInBytes), "walker_table": walker_table(args, run0),
                "texture_mode": int(run0.handle.info().textureMode), "kernel": args.variant, "kernel_launched": kname, "build": build_info(),
                "platform": args.platform, "bytes_total": facts["total_bytes"], "slices_per_rank": len(sset.runs), "matches": facts["matches"],
                "bit_exact": facts["bit_exact"], "bit_exact_method": method, "dist_backend": args.dist_backend if use_dist else None,
                "ranks_seen": sorted(int(r) for r in allf[:, 4]), "rank_placement": rank_placement(allf),
                "folded_result": facts["folded_result"], "folded_reference": facts["folded_reference"],
            },
            "roofline": roofline_block(per_launch, run0.n_read, kname),
            "cpu_baseline": None,
        }
        print(json.dumps(out), flush=True)
        if not all_ok:
            log("[bench] RESULT NOT BIT-EXACT")
            rc = 1
    sset.close()
    if use_dist:
        dist.destroy_process_group()
    return rc


def rank_main(args):
    if args.scaling == "strong":
        return strong_rank_main(args)
    got = setup_rank(args)
    if got is None:
        return 2
    rank, world, device, use_dist, barrier, dist = got
    gpu = args.platform == "gpu"
    import torch
    from pfac_amd import sharding

    t_setup = time.perf_counter()
    perf_mode = None if args.perf_mode is None else (0 if args.perf_mode == "dense" else 1)
    run = Run(args, args.workload, perf_mode, args.texture, rank, world, device)
    info = run.info
    log(f"[bench r{rank}] setup {time.perf_counter() - t_setup:.1f}s: {run.cfg.description}; F={info.numOfPatterns} "
        f"states={info.numOfStates} table={info.sizeOfTableInBytes / 1e6:.1f} MB filter=2^{info.filterLog2Bits} bits "
        f"({info.filterBitsSet} set) CUs={info.multiProcessorCount}")

    if args.worker == "pmc":                                   # profiled by rocprofv3: a few launches, nothing else
        run.step()                                             # as in the timed run: the first launch is over (and has voted on the
        torch.cuda.synchronize()                               # handle's walker, PFACX_WALKER_AUTO) before the measured ones are queued
        for _ in range(PMC_LAUNCHES):
            run.step()
        torch.cuda.synchronize()
        run.close()
        return 0

    # ---- correctness gate (outside the timed region) ------------------------------------------
    run.step()
    if gpu:
        torch.cuda.synchronize()
    log(f"[bench r{rank}] first launch done {time.perf_counter() - t_setup:.1f}s")
    ok, method, pos, ids = run.verify(args, rank, world)
    log(f"[bench r{rank}] verified ({ok}) {time.perf_counter() - t_setup:.1f}s")
    count = int(pos.size)
    checksum = sharding.position_checksum(pos, ids, base=rank * run.n)
    if rank == 0 and args.sparse_file:
        np.savez(args.sparse_file, pos=pos, ids=ids, n=run.n, n_read=run.n_read)

    # ---- timed region: W warmup steps, then exactly K steps between barriers ---------------------
    # Host-side checks leave the GPU idle for ~1 s and its clocks drop; the first ~10 launches after
    # that run up to 15 % slower.  SETTLE_STEPS untimed launches precede the W warmup steps.
    kernel_ms, elapsed = run.timed(args.steps, args.warmup, SETTLE_STEPS if gpu else 0, barrier)
    log(f"[bench r{rank}] timed region done {time.perf_counter() - t_setup:.1f}s")
    # the result of the LAST timed launch, checked like the first (a match lost under back-to-back launches at full clocks
    # would otherwise go unnoticed: the zeros come from other waves than the matches)
    if not args.no_verify:
        pos2, ids2 = run.sparse()
        if not (np.array_equal(pos2, pos) and np.array_equal(ids2, ids)):
            log(f"[bench r{rank}] RESULT OF THE LAST TIMED LAUNCH DIFFERS")
            ok = False

    # ---- gather per-rank facts (RCCL: 7 x int64 per rank: count, checksum, exact, ns, rank, PCI bus id, NUMA node) ----------------------------------------
    allf = sharding.all_gather_facts([count, checksum & 0x7FFFFFFFFFFFFFFF, int(ok), int(elapsed * 1e9), rank, *placement_facts(args, device)],
                                     device=device if (gpu and args.dist_backend == "nccl") else None, force=args.force_dist)
    elapsed_max = float(allf[:, 3].max()) / 1e9
    total_matches, folded = sharding.combine_checksums([(int(c), int(s)) for c, s in allf[:, :2]])
    all_ok = bool(allf[:, 2].all())
    ranks_seen = sorted(int(r) for r in allf[:, 4])

    rc = 0
    if rank == 0:
        # the omp_PFAC.cpp:396-439 check: the folded result of all slices == the reference's folded digests
        expected = None
        dgs = [load_digest(run.cfg.name, r, args.size_mib, last=(r == world - 1)) for r in range(world)]
        if all(d is not None for d in dgs):
            ec, es = sharding.combine_checksums([(d["match_count"], d["checksum"] & 0x7FFFFFFFFFFFFFFF) for d in dgs])
            expected = {"match_count": ec, "checksum": es & 0x7FFFFFFFFFFFFFFF,
                        "equal": bool(ec == total_matches and (es & 0x7FFFFFFFFFFFFFFF) == (folded & 0x7FFFFFFFFFFFFFFF))}
            all_ok = all_ok and expected["equal"]
        if ranks_seen != list(range(world)):
            all_ok = False
        n = run.n
        kname = kernel_name(args, run)
        out = {
            "metric": "input GB/s scanned (PFAC_matchFromDevice, bit-exact)" if gpu else
                      "input GB/s scanned (DRY RUN on the CPU_OMP platform, not the metric)",
            "value": round(world * n / (elapsed_max / args.steps) / 1e9, 2), "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "settle_steps": SETTLE_STEPS if gpu else 0,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": f"{run.cfg.name}: {run.cfg.description}; {args.size_mib} MiB per GPU x {world} GPU(s)"
                            + (f", slices overlap {sharding.overlap_bytes(info.maxPatternLen)} B" if world > 1 else ""),
                "patterns": info.numOfPatterns, "states": info.numOfStates,
                "table": "hashed" if run.perf_mode else "dense", "table_bytes": int(info.sizeOfTableInBytes),
                "walker_table": walker_table(args, run),
                "texture_mode": int(run.handle.info().textureMode), "kernel": args.variant, "kernel_launched": kname, "walker_requested": args.walker,
                "prefilter": {"ladder_last_level": int(getattr(info, "filterLadderLast", 20) or 20), "tail_table_entries": int(getattr(info, "filterTailEntries", 0) or 0),
                              "note": "tail_table_entries > 0: the full-result kernel runs its VETO instance (window walker behind the tail-hash veto, DESIGN 3.1)"},
                "build": build_info(), "platform": args.platform,
                "bytes_per_gpu": n, "matches": total_matches, "bit_exact": all_ok, "bit_exact_method": method,
                "ranks_seen": ranks_seen, "rank_placement": rank_placement(allf), "dist_backend": args.dist_backend if use_dist else None,
                "folded_result": {"match_count": total_matches, "checksum": folded & 0x7FFFFFFFFFFFFFFF},
                "folded_reference": expected,
            },
            "roofline": roofline_block(kernel_ms, run.n_read, kname),
        }
        st = run.scan_stats()
        if st:
            out["config"]["walk_stats"] = st
        if world == 1 and gpu:
            torch = run.torch
            # not the headline metric: the compacted-output API on the same buffers.
            # Synchronous (the match count returns to the host), so wall clock per call.
            d_pos = torch.empty_like(run.d_out)
            run.handle.matchFromDeviceReduce(run.d_in.data_ptr(), run.n_read, run.d_out.data_ptr(), d_pos.data_ptr())
            torch.cuda.synchronize()
            from pfac_amd import hiprt
            rev = [(hiprt.Event(), hiprt.Event()) for _ in range(10)]
            walls = []
            for ea, eb in rev:
                t0r = time.perf_counter()
                ea.record(0)
                _, rcount = run.handle.matchFromDeviceReduce(run.d_in.data_ptr(), run.n_read, run.d_out.data_ptr(), d_pos.data_ptr())
                eb.record(0)
                walls.append(time.perf_counter() - t0r)             # the call is synchronous: the count is back
            torch.cuda.synchronize()
            tr = float(np.median(walls))                            # medians of 10 calls: one slow call (a clock dip, a host hiccup) is not the figure
            gpu_ms = float(np.median([ea.elapsed_ms(eb) for ea, eb in rev]))        # scan kernel (the input's ends ride along in it) + ordering launches + the completion word, on the launch stream
            # the scan kernel alone: the library brackets its launch with HIP events when asked (two more calls, not in the figures above)
            run.handle.setKernelTiming(True)
            kms = []
            for _ in range(3):
                run.handle.matchFromDeviceReduce(run.d_in.data_ptr(), run.n_read, run.d_out.data_ptr(), d_pos.data_ptr())
                kms.append(run.handle.scanStats().get("filterKernelMs"))
            run.handle.setKernelTiming(False)
            kernel_ms_reduce = float(np.median([k for k in kms if k is not None])) if any(k is not None for k in kms) else None
            rp = d_pos[:rcount].cpu().numpy().astype(np.int64)
            ri = run.d_out[:rcount].cpu().numpy()
            keep = rp < n
            same = bool(np.array_equal(rp[keep], pos) and np.array_equal(ri[keep], ids))
            algo = int(run.n_read + 8 * rcount)
            out["reduce_api"] = {"value": round(run.n_read / tr / 1e9, 2), "unit": "GB/s", "ms_per_call": round(tr * 1e3, 4), "calls": len(rev), "statistic": "median",
                                 "gpu_ms": round(gpu_ms, 4), "host_overhead_ms": round(tr * 1e3 - gpu_ms, 4),
                                 "kernel_ms": None if kernel_ms_reduce is None else round(kernel_ms_reduce, 4),
                                 "behind_the_kernel_ms": None if kernel_ms_reduce is None else round(gpu_ms - kernel_ms_reduce, 4),
                                 "matches": int(rcount), "same_result_as_full_vector": same,
                                 "pairs_checksum": int(sharding.position_checksum(rp[keep], ri[keep])), "full_vector_checksum": int(checksum),
                                 "algorithmic_bytes_per_call": algo,
                                 "roofline": {"bound": "hbm", "achieved": round(algo / (gpu_ms / 1e3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": round(algo / (gpu_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                                              "note": "N input bytes + 8 B per match over the GPU time of the call (HIP events around the call: scan kernel, four ordering launches that clean up "
                                                      "behind themselves, a one-thread launch that tells the polling host the call is through; the events themselves are recorded by an idle "
                                                      "queue and include its wake-up); kernel_ms = the scan kernel alone (PFACX_setKernelTiming; rocprofv3 kernel stats in profiles/ agree); "
                                                      "this path is bound by the scanning waves' instruction issue, not by HBM"},
                                 "note": "PFAC_matchFromDeviceReduce, synchronous (the match count returns to the host); ~1 B of HBM traffic per input byte"}
            if not same:
                all_ok = False
                out["config"]["bit_exact"] = False
            del d_pos
            # The same launch with the input and the result vector in other allocations: on this hardware the time
            # of a launch depends on WHERE the two buffers were placed (two classes ~8 % apart that follow the pair
            # of allocations, not offsets inside them: DESIGN.md 3.3).  The headline above is the first pair this
            # process allocated; nothing is picked.
            if not args.no_other_configs or args.spread:
                alt_in = torch.from_numpy(run.host_in).to(device)
                alt_out = torch.full_like(run.d_out, -1)
                spread = {}
                for name_in, b_in in (("first", run.d_in), ("second", alt_in)):
                    for name_out, b_out in (("first", run.d_out), ("second", alt_out)):
                        saved = (run.d_in, run.d_out)
                        run.d_in, run.d_out = b_in, b_out
                        ms, _ = run.timed(10, 2, SETTLE_STEPS)      # the check of the previous pair left the GPU idle: same settling as the headline
                        p3, i3 = run.sparse()
                        if not (np.array_equal(p3, pos) and np.array_equal(i3, ids)):
                            all_ok = False
                            out["config"]["bit_exact"] = False
                            log(f"[bench] result differs on buffer pair {name_in}/{name_out}")
                        run.d_in, run.d_out = saved
                        spread[f"input {name_in} / result {name_out} allocation"] = round(float(np.mean(ms)), 4)
                out["roofline"]["placement_spread_kernel_ms"] = spread
                del alt_in, alt_out
            # PCIe-inclusive rate of PFAC_matchFromHost (never the metric): 256 MiB of the same stream
            # from pageable and from pinned host buffers, 5 B per input byte cross the link
            if not args.no_other_configs:
                hn = min(run.n, 256 << 20)
                host_path = {"bytes": hn}
                # PFACX_prepare: staging pieces, ordering scratch, code objects and the runtime's pageable staging ahead of the first call (round 5's
                # driver line: first call 103 ms).  `first_call_ms` below is the first call BEHIND it, on buffers the process has just allocated.
                t0h = time.perf_counter()
                run.handle.prepare(hn)
                host_path["prepare_ms"] = round((time.perf_counter() - t0h) * 1e3, 3)
                for kind in ("pageable", "pinned"):
                    h_in = torch.from_numpy(run.host_in[:hn].copy())
                    h_out = torch.full((hn,), -1, dtype=torch.int32)           # (touched: the first call below pays for the library's first-call work, not for the page faults of a fresh vector)
                    if kind == "pinned":
                        h_in, h_out = h_in.pin_memory(), h_out.pin_memory()
                    # the first calls on fresh buffers pay for page faults and for the runtime's staging of pageable memory (33 ms, then
                    # 12 ms, then steady: PFAC_HOST_TRACE); the steady state is what a stream of calls on reused buffers sees
                    t0h = time.perf_counter()
                    run.handle.matchFromHost(h_in.data_ptr(), hn, h_out.data_ptr())
                    first_ms = (time.perf_counter() - t0h) * 1e3
                    run.handle.matchFromHost(h_in.data_ptr(), hn, h_out.data_ptr())
                    times = []
                    for _ in range(HOST_CALLS):
                        t0h = time.perf_counter()
                        run.handle.matchFromHost(h_in.data_ptr(), hn, h_out.data_ptr())
                        times.append(time.perf_counter() - t0h)
                    th = float(np.median(times))
                    keep = pos < hn - info.maxPatternLen
                    hp = np.flatnonzero(h_out.numpy()[: hn - info.maxPatternLen])
                    host_path[kind] = {"input_GBps": round(hn / th / 1e9, 2), "ms_per_call": round(th * 1e3, 3), "statistic": f"median of {HOST_CALLS} calls after 2",
                                       "p50_ms": round(th * 1e3, 3), "p90_over_p50": round(float(np.percentile(times, 90)) / th, 3),
                                       "best_ms": round(min(times) * 1e3, 3), "p90_ms": round(float(np.percentile(times, 90)) * 1e3, 3),
                                       "worst_ms": round(max(times) * 1e3, 3), "first_call_ms": round(first_ms, 3),
                                       "same_result": bool(np.array_equal(hp, pos[keep]))}
                    if kind == "pinned":
                        # PFAC_matchFromHostReduce through the same pieces (upload of piece i + 1 beside the compacted scan of piece i, the
                        # pairs appended in position order): 1 B per position over the link, no result vector to fill
                        r_ids = torch.empty(hn, dtype=torch.int32).pin_memory()
                        r_pos = torch.empty(hn, dtype=torch.int32).pin_memory()
                        for _ in range(2):
                            _, cnt = run.handle.matchFromHostReduce(h_in.data_ptr(), hn, r_ids.data_ptr(), r_pos.data_ptr())
                        rtimes = []
                        for _ in range(HOST_CALLS):
                            t0h = time.perf_counter()
                            _, cnt = run.handle.matchFromHostReduce(h_in.data_ptr(), hn, r_ids.data_ptr(), r_pos.data_ptr())
                            rtimes.append(time.perf_counter() - t0h)
                        tr = float(np.median(rtimes))
                        kp = pos < hn - info.maxPatternLen
                        got_pos = r_pos.numpy()[:cnt]
                        sel = got_pos < hn - info.maxPatternLen
                        host_path["reduce"] = {"input_GBps": round(hn / tr / 1e9, 2), "ms_per_call": round(tr * 1e3, 3), "statistic": f"median of {HOST_CALLS} calls after 2, pinned buffers",
                                               "p50_ms": round(tr * 1e3, 3), "p90_over_p50": round(float(np.percentile(rtimes, 90)) / tr, 3),
                                               "best_ms": round(min(rtimes) * 1e3, 3), "p90_ms": round(float(np.percentile(rtimes, 90)) * 1e3, 3), "pairs": int(cnt),
                                               "same_result": bool(np.array_equal(got_pos[sel], pos[kp]) and np.array_equal(r_ids.numpy()[:cnt][sel], ids[kp])),
                                               "device_bytes_tables_plus_scratch": int(run.handle.info().deviceTableBytes + run.handle.info().deviceScratchBytes)}
                        del r_ids, r_pos
                    del h_in, h_out
                # what bounds it: the upload (1 B per position over the link) and the zero fill of the caller's vector by
                # the call's helper threads (4 B per position of host memory bandwidth)
                probe_in = torch.from_numpy(run.host_in[:hn].copy()).pin_memory()
                torch.cuda.synchronize()
                t0h = time.perf_counter()
                for _ in range(3):
                    run.d_in[:hn].copy_(probe_in, non_blocking=True)
                torch.cuda.synchronize()
                host_path["link_h2d_GBps_pinned"] = round(3 * hn / (time.perf_counter() - t0h) / 1e9, 1)
                fill = np.empty(hn, dtype=np.int32)
                fill.fill(1)
                t0h = time.perf_counter()
                fill.fill(0)
                host_path["host_zero_fill_GBps_one_thread"] = round(4 * hn / (time.perf_counter() - t0h) / 1e9, 1)
                host_path["bound"] = "per position: 1 B over the host link + 4 B of zero fill in host memory (2-8 helper threads, streaming stores, piece by piece) " \
                                     "+ the pairs of the matches, scattered piece by piece; the uploads are queued by a thread of their own, the scan of a " \
                                     "32 Mi-position piece runs beside the upload of the next.  Fill and link share the host's memory channels."
                host_path["of_the_link"] = {k: round(host_path[k]["input_GBps"] / host_path["link_h2d_GBps_pinned"], 3) for k in ("pageable", "pinned", "reduce")}
                del probe_in, fill
                out["host_path_pcie_inclusive"] = host_path
            # What this part sustains for the traffic shape of the path with nothing else in it ("also
            # measure a 1R:4W streaming kernel as the achievable ceiling and report both"): pfac_stream_1r4w of the
            # kernel module, on the very buffers just timed.  It zeroes d_out: everything above has been checked.
            try:
                import ctypes as C
                from pfac_amd import api as _api
                mod = C.CDLL(_api.library_paths()[1])
                mod.PFACX_streamProbe.restype = C.c_double
                mod.PFACX_streamProbe.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
                n4k = run.n_read // 4096 * 4096
                sms = float(mod.PFACX_streamProbe(run.d_in.data_ptr(), run.d_out.data_ptr(), n4k, 10))
                if sms > 0:
                    out["roofline"]["bare_stream_1r4w"] = {
                        "ms": round(sms, 4), "frac": round(5.0 * n4k / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "scan_over_stream": round(out["roofline"]["kernel_ms_avg"] * n4k / run.n_read / sms, 3),
                        "note": "every wave reads 1 KiB and writes 4 KiB of zeros, same buffers, 10 launches"}
            except (OSError, AttributeError) as e:
                log(f"[bench] stream probe skipped: {e}")
            if not args.no_other_configs:
                buffers = (run.d_in, run.d_out)
                if args.variant == "auto":
                    out["call_size_sweep"] = size_sweep(run)
                    log(f"[bench] call sizes: { {k: v.get('input_GBps', v.get('cold_over_warm')) for k, v in out['call_size_sweep'].items()} }")
                out["other_configs"] = other_configs(args, device, buffers)
                if args.workload == "c3" and args.size_mib == 1024 and args.perf_mode is None:
                    # BASELINE config 4 (8 GiB) as ONE workload on this one GPU: what `--scaling strong --gpus 1` reports, the
                    # N = 1 point of the 1/2/4/8 strong-scaling curve
                    t0 = time.perf_counter()
                    sset = SliceSet(args, "c3", None, args.texture, 8, 0, 1, device)
                    ok4, method4, count4, sum4 = sset.gate(args)
                    ms4, el4 = sset.timed(3, 1, 4)
                    out["other_configs"]["c4_8gib_one_gpu"], _ = strong_numbers(args, sset, ms4, el4, count4, sum4, ok4, method4, 1)
                    sset.close()
                    log(f"[bench] c4_8gib_one_gpu: {out['other_configs']['c4_8gib_one_gpu']['aggregate_GBps']} GB/s ({time.perf_counter() - t0:.1f}s)")
                if not all(e["bit_exact"] for e in out["other_configs"].values()):
                    all_ok = False
        out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
        if not all_ok:
            log("[bench] RESULT NOT BIT-EXACT")
            rc = 1
    run.close()
    if use_dist:
        dist.destroy_process_group()
    return rc


# ---------------------------------------------------------------------------------- orchestrator

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def orchestrate(args, argv):
    """No GPU call, no torch import in this process: children are started fresh (never exec'ed into)."""
    p = subprocess.run([sys.executable, "-c", "import __graft_entry__ as e; e.build(only_if_missing=True)"], cwd=ROOT)
    if p.returncode != 0:
        log("[bench] build failed")
        return 2
    os.makedirs(SCRATCH, exist_ok=True)
    n = args.gpus
    single = n == 1 and args.platform == "gpu" and args.scaling == "weak"      # the CPU baseline and the PMC passes belong to the headline (weak) line
    sparse_file = os.path.join(SCRATCH, f"{args.workload}_rank0_sparse.npz")
    if os.path.exists(sparse_file):
        os.remove(sparse_file)
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        extra = ["--sparse-file", sparse_file] if (single and r == 0 and not args.no_cpu_baseline) else []
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + extra, env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    rc = max((abs(c) for c in rcs), default=0)
    line = None
    for l in out0.decode(errors="replace").splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
        else:
            log(l)
    if line is None:
        log(f"[bench] rank 0 printed no result (exit codes {rcs})")
        return rc or 1
    out = json.loads(line)

    if single and not args.no_cpu_baseline:
        threads = lscpu_facts()["physical_cores"]
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", "cpu", "--workload", args.workload,
               "--size-mib", str(args.size_mib), "--cpu-seconds", str(args.cpu_seconds), "--sparse-file", sparse_file]
        if args.perf_mode:
            cmd += ["--perf-mode", args.perf_mode]
        p = subprocess.run(cmd, cwd=ROOT, env=cpu_worker_env(threads), stdout=subprocess.PIPE)
        try:
            out["cpu_baseline"] = json.loads(p.stdout.decode().strip().splitlines()[-1])
            if out["cpu_baseline"].get("gpu_result_equals_cpu_result") is False:
                out["config"]["bit_exact"] = False
                rc = rc or 1
        except Exception:
            log(f"[bench] CPU baseline worker failed (rc {p.returncode})")
    if single and args.pmc == "auto" and args.variant not in ("naive", "reftable"):
        common = ["--workload", args.workload, "--size-mib", str(args.size_mib), "--texture", args.texture, "--no-verify"]
        if args.perf_mode:
            common += ["--perf-mode", args.perf_mode]
        t0 = time.perf_counter()
        traffic, src = measure_traffic(common, "pfac_scan_filter")
        log(f"[bench] PMC passes {time.perf_counter() - t0:.1f}s: {traffic} ({src})")
        if traffic:
            out["roofline"]["traffic"] = traffic["bytes"]
            out["roofline"]["traffic_source"] = src
            out["roofline"]["traffic_detail"] = traffic
        else:
            out["roofline"]["traffic_note"] = src
    if single and args.pmc == "auto" and args.variant not in ("naive", "reftable") and out["roofline"]["traffic"] is None \
            and args.size_mib == 1024 and args.perf_mode is None:      # the PMC passes of this run failed: last committed profile
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = committed_traffic(args.workload, out["roofline"]["kernel"])
    print(json.dumps(out), flush=True)
    return rc


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.worker == "cpu":
        return cpu_worker(args)
    if args.worker in ("pmc", "rank") or "RANK" in os.environ:
        return rank_main(args)
    return orchestrate(args, argv)


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""bench.py -- input GB/s scanned by the PFAC match path on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c5] [--size-mib 1024]

One "step" = one PFAC_matchFromDevice() pass over the rank's slice of the synthetic stream,
input and result buffers resident in HBM, transition tables already uploaded.  N > 1 is
launched by torch.distributed.run (one process per GPU); the stream is sharded as independent
1 GiB slices with a maxPatternLen+1 read-ahead tail (reference omp_PFAC.cpp:319-377), no
data-path collective, RCCL only gathers the per-rank (match count, checksum) pairs.

Rank 0 prints ONE JSON line (the driver contract) that also carries
  "roofline":     algorithmic HBM bytes (5 B per input byte: 1 read + 4 written) / kernel time,
                  kernel time measured with HIP events on the launch stream inside the timed region
  "cpu_baseline": the reference's own OpenMP matcher (oracle/_ref, compiled from the unmodified
                  reference sources) timed on this host's cores on a bounded sample, N=1 only.
"""

import argparse
import json
import os
import sys
import time

# OpenMP workers (numpy/torch pools, the oracle's cross-check before the timed region) must sleep, not
# spin, once their parallel region is over: spinning host threads delay the launches of the timed
# steps.  Has to be in the environment before the first OpenMP runtime is loaded.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
ALGO_BYTES_PER_INPUT_BYTE = 5    # 1 B input read + 4 B int32 result written
SETTLE_STEPS = 32                # untimed launches after the host-side cross-check, before the warmup steps


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c5"])
    ap.add_argument("--size-mib", type=int, default=1024, help="input bytes per GPU, MiB")
    ap.add_argument("--variant", default="filter", choices=["filter", "naive"])
    ap.add_argument("--texture", default="auto", choices=["auto", "on", "off"])
    ap.add_argument("--perf-mode", default=None, choices=[None, "dense", "hash"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) is the real path; gloo lets several ranks share one GPU for a dry run")
    return ap.parse_args()


def sparse_result(d_out, n):
    """(positions, ids) of the non-zero results of a device int32 result vector."""
    import torch
    pos = torch.nonzero(d_out[:n]).flatten()
    ids = d_out[:n][pos]
    return pos.cpu().numpy().astype(np.int64), ids.cpu().numpy().astype(np.int64)


def verify(handle, api, ob, cfg_pattern_file, host_in, d_in, d_out, n, n_read, perf_mode, variant):
    """Bit-exactness outside the timed region:
       (1) full size: the timed kernel's result == the other, independent kernel's result;
       (2) sampled 1 MiB windows of the stream re-scanned by the oracle == the same windows of (1)."""
    import torch
    torch.cuda.synchronize()
    pos, ids = sparse_result(d_out, n)
    # (1) second, independent kernel on the same device buffers
    d_chk = torch.full_like(d_out, -1)
    other = api.PFACX_KERNEL_NAIVE if variant == api.PFACX_KERNEL_FILTER else api.PFACX_KERNEL_FILTER
    handle.setKernelVariant(other)
    handle.matchFromDevice(d_in.data_ptr(), n_read, d_chk.data_ptr())
    handle.setKernelVariant(variant)
    torch.cuda.synchronize()
    same = bool(torch.equal(d_out[:n_read], d_chk[:n_read]))
    del d_chk
    # (2) oracle on windows (window + overlap so matches crossing the window end are exact)
    oracle = ob.Oracle(cfg_pattern_file, dense=(perf_mode == 0), hashed=(perf_mode == 1))
    rng = np.random.Generator(np.random.PCG64(99))
    win = 1 << 20
    tail = oracle.max_pattern_len + 1
    starts = [0, max(0, n - win)] + [int(x) for x in rng.integers(0, max(1, n - win), size=6)]
    windows_ok = True
    for s in starts:
        e = min(n, s + win)
        r = min(n_read, e + tail)
        want = oracle.match(host_in[s:r], hashed=(perf_mode == 1), omp=True)[: e - s]
        got = d_out[s:e].cpu().numpy()
        if not np.array_equal(got, want):
            windows_ok = False
            log(f"[verify] MISMATCH in window [{s},{e})")
    oracle.close()
    return same and windows_ok, pos, ids


def committed_traffic(workload, kernel_substr):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh ->
    profiles/rNN_hbm_traffic_<workload>.json): 2 x FETCH_SIZE (gfx950 reports half of coalesced
    reads -- calibrated on the stream probe in the same file) + WRITE_SIZE, in bytes.  None if no
    profile of this workload is committed.  PMC passes cannot run inside the timed bench process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_hbm_traffic_{workload}.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        # pfac_scan_filter<MODE, HAS_SHORT, REDUCE>: the full-result kernel is the REDUCE = false instance
        def mine(k):
            return kernel_substr in k and not k.rstrip(">( ").endswith("true")
        f = [v for k, v in d["FETCH_SIZE"]["scan"].items() if mine(k)]
        w = [v for k, v in d["WRITE_SIZE"]["scan"].items() if mine(k)]
        if not f or not w:
            return None, None
        return int((2.0 * f[0] + w[0]) * 1024), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def cpu_baseline(ob, pattern_file, host_in, perf_mode, target_seconds):
    """Reference OpenMP matcher (oracle/_ref) -- or the C port if _ref is absent -- on a bounded sample."""
    threads = ob.omp_max_threads()
    oracle = ob.Oracle(pattern_file, dense=(perf_mode == 0), hashed=(perf_mode == 1))
    use_ref = ob.have_reference()
    if perf_mode == 0:
        dense = oracle.dense_table()
    else:
        row, val = oracle.hash_row(), oracle.hash_val()

    def run(sample):
        t0 = time.perf_counter()
        if use_ref:
            if perf_mode == 0:
                ob.Reference.match_dense(sample, dense, oracle.num_patterns, oracle.initial_state, omp=True)
            else:
                ob.Reference.match_hash(sample, row, val, oracle.num_patterns, oracle.initial_state, omp=True)
        else:
            oracle.match(sample, hashed=(perf_mode == 1), omp=True)
        return time.perf_counter() - t0

    pilot = min(host_in.size, 16 << 20)
    t = run(host_in[:pilot])
    rate = pilot / t
    sample = int(min(host_in.size, max(pilot, rate * target_seconds)))
    sample -= sample % (1 << 20) if sample > (1 << 20) else 0
    t = run(host_in[:sample])
    oracle.close()
    return {
        "value": round(sample / t / 1e9, 4), "unit": "GB/s", "cores": threads,
        "kind": "reference" if use_ref else "port",
        "sample": f"first {sample >> 20} MiB of the rank-0 stream, {'PFAC_CPU_OMP_spaceDriven' if perf_mode else 'PFAC_CPU_OMP_timeDriven'}"
                  f" ({'reference sources compiled unmodified' if use_ref else 'oracle C port'}), {threads} OpenMP threads, 1 run after a 16 MiB pilot",
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")

    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU: the match path has no CPU fallback"
    if args.dist_backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()      # dry run: ranks may share a device
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import __graft_entry__ as entry
    if rank == 0:
        entry.build(only_if_missing=True)
    if world > 1:
        dist.barrier()

    from pfac_amd import api, hiprt, sharding
    from pfac_amd import workloads as wl

    # ---- workload ---------------------------------------------------------------------------
    t_setup = time.perf_counter()
    cfg = wl.make_config(args.workload)
    perf_mode = cfg.perf_mode if args.perf_mode is None else (0 if args.perf_mode == "dense" else 1)
    tmp = os.path.join(ROOT, "gpurun_out", "bench")
    os.makedirs(tmp, exist_ok=True)
    pattern_file = os.path.join(tmp, f"{cfg.name}_rank{rank}.pat")
    wl.write_pattern_file(pattern_file, cfg.patterns)

    handle = api.PFAC.create()
    handle.setPerfMode(perf_mode)
    handle.setTextureMode({"auto": api.PFAC_AUTOMATIC, "on": api.PFAC_TEXTURE_ON, "off": api.PFAC_TEXTURE_OFF}[args.texture])
    handle.setKernelVariant(api.PFACX_KERNEL_FILTER if args.variant == "filter" else api.PFACX_KERNEL_NAIVE)
    handle.readPatternFromFile(pattern_file)
    info = handle.info()

    # slice `rank` of the N x size stream plus the head of the next slice (generators are prefix-stable)
    host_in, n = sharding.rank_input(cfg, args.size_mib << 20, rank, world, info.maxPatternLen)
    n_read = host_in.size
    d_in = torch.from_numpy(host_in).to(f"cuda:{local_rank}")
    d_out = torch.full((n_read,), -1, dtype=torch.int32, device=f"cuda:{local_rank}")
    torch.cuda.synchronize()
    log(f"[bench r{rank}] setup {time.perf_counter() - t_setup:.1f}s: {cfg.description}; F={info.numOfPatterns} "
        f"states={info.numOfStates} table={info.sizeOfTableInBytes / 1e6:.1f} MB filter=2^{info.filterLog2Bits} bits "
        f"({info.filterBitsSet} set) CUs={info.multiProcessorCount}")

    def step():
        handle.matchFromDevice(d_in.data_ptr(), n_read, d_out.data_ptr())

    # ---- correctness gate (outside the timed region) ------------------------------------------
    step()
    ok = True
    pos = ids = None
    if not args.no_verify:
        from oracle import binding as ob   # checker only
        ok, pos, ids = verify(handle, api, ob, pattern_file, host_in, d_in, d_out, n, n_read, perf_mode,
                              api.PFACX_KERNEL_FILTER if args.variant == "filter" else api.PFACX_KERNEL_NAIVE)
    else:
        pos, ids = sparse_result(d_out, n)
    count = int(pos.size)
    checksum = sharding.position_checksum(pos, ids, base=rank * n)

    # ---- timed region ---------------------------------------------------------------------------
    events = [(hiprt.Event(), hiprt.Event()) for _ in range(args.steps)]
    # The cross-check above leaves the GPU idle for ~1 s of host work and its clocks drop; the first
    # ~10 launches after that run up to 15 % slower (profiles/: kernel_ms_steps).  SETTLE_STEPS untimed
    # launches bring the clocks back before the W warmup steps the caller asked for.
    for _ in range(SETTLE_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in events:
        a.record(0)
        step()
        b.record(0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    kernel_ms = [a.elapsed_ms(b) for a, b in events]
    kernel_avg_s = float(np.mean(kernel_ms)) / 1e3

    # ---- gather per-rank facts (RCCL: 4 x int64 per rank) ----------------------------------------
    allf = sharding.all_gather_facts([count, checksum & 0x7FFFFFFFFFFFFFFF, int(ok), int(elapsed * 1e9)],
                                     device=f"cuda:{local_rank}" if args.dist_backend == "nccl" else None)
    elapsed_max = float(allf[:, 3].max()) / 1e9
    total_matches = int(allf[:, 0].sum())
    all_ok = bool(allf[:, 2].all())

    if rank == 0:
        ms_per_step = elapsed_max / args.steps * 1e3
        value = world * n / (elapsed_max / args.steps) / 1e9
        achieved = ALGO_BYTES_PER_INPUT_BYTE * n_read / kernel_avg_s / 1e9
        kname = "pfac_scan_filter" if args.variant == "filter" else "pfac_scan_naive"
        traffic, traffic_src = (committed_traffic(cfg.name, kname) if args.size_mib == 1024 and perf_mode == cfg.perf_mode
                                else (None, None))
        out = {
            "metric": "input GB/s scanned (PFAC_matchFromDevice, bit-exact)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": SETTLE_STEPS,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": f"{cfg.name}: {cfg.description}; {args.size_mib} MiB per GPU x {world} GPU(s)"
                            + (f", slices overlap {sharding.overlap_bytes(info.maxPatternLen)} B" if world > 1 else ""),
                "patterns": info.numOfPatterns, "states": info.numOfStates,
                "table": "hashed" if perf_mode else "dense", "table_bytes": int(info.sizeOfTableInBytes),
                "texture_mode": int(handle.info().textureMode), "kernel": args.variant,
                "bytes_per_gpu": n, "matches": total_matches, "bit_exact": all_ok,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": kname,
                "kernel_ms_avg": round(kernel_avg_s * 1e3, 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
                "kernel_ms_median": round(float(np.median(kernel_ms)), 4), "kernel_ms_max": round(float(np.max(kernel_ms)), 4),
                "kernel_ms_steps": [round(float(x), 3) for x in kernel_ms] if len(kernel_ms) <= 64 else None,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_INPUT_BYTE * n_read,
                "input_only_frac": round(n_read / kernel_avg_s / 1e9 / HBM_PEAK_GBS, 4),
            },
        }
        if world == 1:
            # not the headline metric: the compacted-output API on the same buffers.
            # Synchronous (the match count returns to the host), so wall clock per call.
            d_pos = torch.empty_like(d_out)
            handle.matchFromDeviceReduce(d_in.data_ptr(), n_read, d_out.data_ptr(), d_pos.data_ptr())
            torch.cuda.synchronize()
            t0r = time.perf_counter()
            for _ in range(5):
                _, rcount = handle.matchFromDeviceReduce(d_in.data_ptr(), n_read, d_out.data_ptr(), d_pos.data_ptr())
            torch.cuda.synchronize()
            tr = (time.perf_counter() - t0r) / 5
            out["reduce_api"] = {"value": round(n_read / tr / 1e9, 2), "unit": "GB/s", "ms_per_call": round(tr * 1e3, 4),
                                 "matches": int(rcount), "same_matches_as_full_result": bool(rcount == count),
                                 "algorithmic_bytes_per_call": int(n_read + 8 * rcount),
                                 "note": "PFAC_matchFromDeviceReduce incl. count readback and position sort; ~1 B/input byte of HBM traffic"}
            del d_pos
        if world == 1 and not args.no_cpu_baseline:
            from oracle import binding as ob   # cpu_baseline leg
            out["cpu_baseline"] = cpu_baseline(ob, pattern_file, host_in[:n], perf_mode, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
        if not all_ok:
            log("[bench] RESULT NOT BIT-EXACT")

    handle.destroy()
    if world > 1:
        dist.destroy_process_group()
    return 0 if all_ok else 1


if __name__ == "__main__":
    sys.exit(main())

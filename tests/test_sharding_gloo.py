"""N > 1 path on CPU: world_size 2, gloo backend.

Same host logic bench.py uses for --gpus N (slice layout with maxPatternLen+1 overlap, no data-path
collective, all-gather of per-rank (count, checksum) facts), with the match itself done by the
library's PFAC_PLATFORM_CPU_OMP path through the C ABI.  The combined facts must equal those of one
scan over the concatenated stream -- the reference's own multi-GPU self-check
(PFAC/test/omp_PFAC.cpp:396-439) in distributed form.
"""

import os
import socket

import numpy as np
import pytest

from pfac_amd import sharding


def test_plan_slices_cover_and_overlap():
    for total, parts, ov in [(10_000, 3, 61), (1 << 20, 8, 33), (5, 4, 2), (1024, 1, 9), (0, 2, 3)]:
        sl = sharding.plan_slices(total, parts, ov)
        assert sum(s.end - s.start for s in sl) == total
        pos = 0
        for s in sl:
            assert s.start == pos and s.end > s.start and s.read_end == min(s.end + ov, total)
            pos = s.end
        assert pos == total
    assert sharding.rank_slices(7, 1, 3) == [1, 4]
    assert sharding.overlap_bytes(60) == 61      # omp_PFAC.cpp:324


def test_position_checksum_is_additive():
    rng = np.random.Generator(np.random.PCG64(3))
    pos = np.sort(rng.choice(1 << 30, size=1000, replace=False))
    ids = rng.integers(1, 30000, size=1000)
    whole = sharding.position_checksum(pos, ids)
    cut = 400
    parts = [(cut, sharding.position_checksum(pos[:cut], ids[:cut])),
             (1000 - cut, sharding.position_checksum(pos[cut:] - pos[cut], ids[cut:], base=int(pos[cut])))]
    assert sharding.combine_checksums(parts) == (1000, whole)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="2")
    import torch.distributed as dist
    from pfac_amd import api
    from pfac_amd import workloads as wl
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = wl.make_config("c2")
        pf = wl.write_pattern_file(os.path.join(tmp, f"r{rank}.pat"), cfg.patterns)
        h = api.PFAC.createHostOnly()
        h.setPlatform(api.PFAC_PLATFORM_CPU_OMP)
        h.setPerfMode(api.PFAC_SPACE_DRIVEN)
        h.readPatternFromFile(pf)
        data, owned = sharding.rank_input(cfg, n, rank, world, h.info().maxPatternLen)
        # plant patterns: one inside the slice, one straddling the boundary to the next slice
        p0 = np.frombuffer(cfg.patterns[rank], dtype=np.uint8)
        data[100:100 + p0.size] = p0
        pb = np.frombuffer(cfg.patterns[7], dtype=np.uint8)          # starts 3 bytes before the slice boundary
        if rank == 0:
            data[n - 3:n - 3 + pb.size] = pb                         # tail reaches into the overlap
        else:
            data[0:pb.size - 3] = pb[3:]                             # the same bytes seen from slice 1
        res = h.match_host_array(data)[:owned]
        h.destroy()
        pos = np.nonzero(res)[0]
        facts = sharding.all_gather_facts([pos.size, sharding.position_checksum(pos, res[pos], base=rank * n) & 0x7FFFFFFFFFFFFFFF])
        if rank == 0:
            q.put(facts.tolist())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_scan_equals_single_scan(tmp_path):
    import torch.multiprocessing as mp
    from pfac_amd import api
    from pfac_amd import workloads as wl
    world, n = 2, 1 << 18
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    facts = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # single scan over the concatenated stream (with the same planted patterns)
    cfg = wl.make_config("c2")
    pf = wl.write_pattern_file(str(tmp_path / "all.pat"), cfg.patterns)
    whole = np.concatenate([cfg.input_slice(n, r) for r in range(world)])
    for r in range(world):
        p0 = np.frombuffer(cfg.patterns[r], dtype=np.uint8)
        whole[r * n + 100: r * n + 100 + p0.size] = p0
    pb = np.frombuffer(cfg.patterns[7], dtype=np.uint8)
    whole[n - 3:n - 3 + pb.size] = pb
    h = api.PFAC.createHostOnly()
    h.readPatternFromFile(pf)
    res = h.match_host_array(whole)
    h.destroy()
    pos = np.nonzero(res)[0]
    want = (pos.size, sharding.position_checksum(pos, res[pos]))
    got = sharding.combine_checksums([(c, s) for c, s in facts])
    assert pos.size >= world + 1 and res[n - 3] == 8, "the boundary-straddling pattern must be found"
    assert got[0] == want[0]
    assert got[1] == (sum(s for _, s in facts) & 0xFFFFFFFFFFFFFFFF)
    # per-rank checksums were masked to 63 bits for the int64 transport: compare in that domain
    per_rank = []
    for r in range(world):
        sel = (pos >= r * n) & (pos < (r + 1) * n)
        per_rank.append([int(sel.sum()), sharding.position_checksum(pos[sel], res[pos[sel]]) & 0x7FFFFFFFFFFFFFFF])
    assert facts == per_rank


def test_bench_spawns_the_ranks_itself_and_checks_the_reference_digests():
    """`python bench.py --gpus 2` must start two rank processes on its own (the driver's command shape) and fold
    their results against the committed digests of the reference's CPU/OMP output (omp_PFAC.cpp:396-439).  Here on
    the library's CPU_OMP platform over gloo (no GPU in this container): slices 0 and 1 of the 8 MiB c3 stream."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--platform", "cpu_omp", "--dist-backend", "gloo",
                        "--size-mib", "8", "--steps", "1", "--warmup", "0", "--workload", "c3"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["ranks_seen"] == [0, 1]
    assert out["config"]["bit_exact"] is True and "digest" in out["config"]["bit_exact_method"]
    assert out["config"]["folded_reference"]["equal"] is True
    assert out["config"]["folded_result"]["match_count"] == 4579 + 4590      # tests/golden/full_digests.json, slices 0 and 1
    # a rank count that does not match --gpus is refused, not silently reported
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--platform", "cpu_omp", "--dist-backend", "gloo",
                        "--size-mib", "8", "--steps", "1", "--warmup", "0"], cwd=root, env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 2 and b"refusing" in p.stderr


def test_strong_scaling_mode_deals_the_slices_round_robin():
    """`bench.py --scaling strong`: BASELINE config 4 as ONE workload -- the stream's 8 slices dealt round-robin over the
    ranks (rank r scans slices r, r + N, ...: PFAC/test/omp_PFAC.cpp:351-355, pfac_amd/sharding.py rank_slices), the same total
    work at every N.  Two ranks x four slices of the 8 x 8 MiB c3 stream on the CPU_OMP platform over gloo: every slice
    equals its reference digest, the folded result equals the folded digests of all eight; and the same stream on one rank
    gives the same folded result."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    folded = {}
    for gpus in (2, 1):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--scaling", "strong", "--total-mib", "64", "--size-mib", "8",
                            "--platform", "cpu_omp", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0", "--workload", "c3"],
                           cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        out = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
        assert out["scaling"] == "strong" and out["n_gpus"] == gpus and out["config"]["slices_per_rank"] == 8 // gpus
        assert out["config"]["bytes_total"] == 64 << 20
        assert out["config"]["bit_exact"] is True and out["config"]["folded_reference"]["equal"] is True
        assert out["config"]["folded_result"]["match_count"] == 4579 + 4590 + 4579 + 4571 + 4454 + 4655 + 4583 + 4493   # tests/golden/full_digests.json
        folded[gpus] = out["config"]["folded_result"]
    assert folded[1] == folded[2]
    assert sharding.rank_slices(8, 1, 2) == [1, 3, 5, 7]


@pytest.mark.timeout(900)
def test_eight_ranks_one_slice_each_fold_to_the_reference():
    """The rank count the driver's scaling run ends at: `bench.py --gpus 8 --scaling strong` at toy size -- eight gloo ranks on the
    CPU_OMP platform, one 8 MiB slice of the c3 stream each (rank r scans slice r with the head of slice r + 1:
    PFAC/test/omp_PFAC.cpp:324,351-355).  Every slice equals its reference digest, rank 0 folds eight (count, checksum) pairs
    into the folded digests of all eight slices and sees every rank; a world of eight under --gpus 4 is refused."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--scaling", "strong", "--total-mib", "64", "--size-mib", "8",
                        "--platform", "cpu_omp", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0", "--workload", "c3"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=850)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["scaling"] == "strong" and out["n_gpus"] == 8 and out["config"]["slices_per_rank"] == 1
    assert out["config"]["ranks_seen"] == list(range(8))
    assert out["config"]["bit_exact"] is True and out["config"]["folded_reference"]["equal"] is True
    assert out["config"]["folded_result"]["match_count"] == 4579 + 4590 + 4579 + 4571 + 4454 + 4655 + 4583 + 4493   # tests/golden/full_digests.json
    env2 = dict(env, RANK="0", WORLD_SIZE="8", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--platform", "cpu_omp", "--dist-backend", "gloo",
                        "--size-mib", "8", "--steps", "1", "--warmup", "0"], cwd=root, env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 2 and b"refusing" in p.stderr

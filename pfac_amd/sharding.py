"""Slice planning for multi-GPU / chunked scans.

Mirrors the host logic of the reference's multi-GPU example
(``PFAC/test/omp_PFAC.cpp:319-377``): every start position is independent, the
only coupling between neighbouring slices is that a walk may read up to
``maxPatternLen - 1`` bytes past its slice, so a slice is scanned together with
a tail of ``maxPatternLen + 1`` bytes of its successor (the reference's guard,
``omp_PFAC.cpp:324,353``) and only the results for ``[start, end)`` are kept
(``omp_PFAC.cpp:377``).  No data-path collective is needed; ranks exchange only
``(match_count, checksum)`` pairs.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import List


def overlap_bytes(max_pattern_len: int) -> int:
    """Tail a slice must see beyond its end (ref omp_PFAC.cpp:324: max_patternLen + 1)."""
    return int(max_pattern_len) + 1


@dataclass(frozen=True)
class Slice:
    index: int
    start: int      # first position this slice owns
    end: int        # one past the last position it owns
    read_end: int   # one past the last input byte it must read (end + overlap, clamped)


def plan_slices(total: int, parts: int, overlap: int, align: int = 1024) -> List[Slice]:
    """Split ``[0, total)`` into ``parts`` contiguous owned ranges with ``overlap`` read-ahead.

    Boundaries are rounded down to ``align`` (the kernel's tile size) so every slice but the
    last is a whole number of tiles; empty slices are dropped.
    """
    assert parts >= 1 and total >= 0 and overlap >= 0
    bounds = [0]
    for i in range(1, parts):
        b = (total * i // parts) // align * align
        bounds.append(max(b, bounds[-1]))
    bounds.append(total)
    out = []
    for i in range(parts):
        s, e = bounds[i], bounds[i + 1]
        if e > s:
            out.append(Slice(len(out), s, e, min(e + overlap, total)))
    return out


def rank_slices(total_slices: int, rank: int, world: int) -> List[int]:
    """Static round-robin of slice indices over ranks (ref omp_PFAC.cpp:351: tid + k*num_threads)."""
    return list(range(rank, total_slices, world))


def combine_checksums(parts):
    """Fold per-slice (count, checksum) pairs, in slice order, into one pair."""
    count, acc = 0, 0
    for c, s in parts:
        count += int(c)
        acc = (acc + int(s)) & 0xFFFFFFFFFFFFFFFF
    return count, acc


def position_checksum(positions, ids, base: int = 0) -> int:
    """Order-independent 64-bit checksum of a sparse result: sum over matches of
    mix(global position) * id  (mod 2^64).  Additive, so per-slice values simply add up."""
    import numpy as np
    p = (np.asarray(positions, dtype=np.uint64) + np.uint64(base))
    v = np.asarray(ids, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = p * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)
        h ^= h >> np.uint64(29)
        return int((h * v).sum(dtype=np.uint64))


def rank_input(cfg, n: int, rank: int, world: int, max_pattern_len: int):
    """Input one rank scans in the weak-scaling layout used by bench.py (BASELINE config 4): the
    global stream is `world` slices of n bytes, slice i generated from seed+i; every rank but the
    last appends the first overlap_bytes() bytes of its successor (generators are prefix-stable).
    Returns (host_array, owned_bytes)."""
    import numpy as np
    overlap = overlap_bytes(max_pattern_len) if rank < world - 1 else 0
    buf = np.empty(n + overlap, dtype=np.uint8)
    buf[:n] = cfg.input_slice(n, rank)
    if overlap:
        buf[n:] = cfg.input_slice(overlap, rank + 1)
    return buf, n


def all_gather_facts(values, device=None, force=False):
    """All-gather a short list of int64 facts from every rank (RCCL on GPU, gloo on CPU).
    Returns a [world, len(values)] numpy array; no-op for a single process unless `force` (then the collective runs
    with a world of one: the same RCCL code path on a machine with one GPU)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return t.cpu().numpy()[None, :]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()

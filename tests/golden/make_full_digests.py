#!/usr/bin/env python3
"""Generate tests/golden/full_digests.json: digests of the REFERENCE's own OpenMP matchers
(PFAC/src/PFAC_CPU_OMP.cpp:81-185, compiled unmodified into oracle/_ref) over the full-size
BASELINE.json streams.

Runs only in the build container (needs oracle/_ref, i.e. /root/reference).  Only data is written:
seeds, sizes, fingerprints of the generated inputs and, per result vector,

    match_count   number of non-zero results
    checksum      pfac_amd.sharding.position_checksum (additive over slices, base = slice * n)
    fnv1a64       FNV-1a-64 over the int32 little-endian result vector
    sha256        SHA-256 over the same bytes

A slice of the multi-GPU stream (BASELINE config 4, workload c3) is scanned together with the first
maxPatternLen + 1 bytes of its successor (omp_PFAC.cpp:324) -- variant "inner" -- unless it is the last
slice of the job -- variant "last", where the stream ends with the slice.  Both are recorded for every
slice so that any world size 1..8 can be checked.

    python tests/golden/make_full_digests.py [--size-mib 1024] [--only c2,c3,c5,c6]
"""

import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as ob               # noqa: E402
from pfac_amd import sharding                   # noqa: E402
from pfac_amd import workloads as wl            # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "full_digests.json")
SLICES = {"c2": 1, "c3": 8, "c5": 1, "c6": 1}
HASHED_ONLY = ("c3", "c6")        # Snort-scale sets: the dense table is S KiB = 0.5 GB


def digest(result, base):
    pos = np.flatnonzero(result)
    ids = result[pos]
    fnv, cnt = ob.digest(result)
    assert cnt == pos.size
    return {
        "match_count": int(pos.size),
        "checksum": int(sharding.position_checksum(pos, ids, base=base)),
        "fnv1a64": int(fnv),
        "sha256": hashlib.sha256(result.view(np.uint8)).hexdigest(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size-mib", type=int, default=1024)
    ap.add_argument("--only", default="c2,c3,c5")
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    assert ob.have_reference(), "build oracle/_ref first: make -C oracle ref"
    n = args.size_mib << 20
    doc = json.load(open(args.out)) if os.path.exists(args.out) else {
        "_doc": "digests of the reference's PFAC_CPU_OMP_* (oracle/_ref) over the full-size streams; see make_full_digests.py",
        "records": []}
    tmp = tempfile.mkdtemp()
    for name in args.only.split(","):
        cfg = wl.make_config(name)
        pf = wl.write_pattern_file(os.path.join(tmp, name + ".pat"), cfg.patterns)
        ref = ob.Reference(pf)
        o = ob.Oracle(pf, dense=(name not in HASHED_ONLY), hashed=True)
        assert ref.edges() == o.edges()
        F, init, maxlen = ref.t.num_patterns, ref.t.initial_state, ref.t.max_pattern_len
        overlap = sharding.overlap_bytes(maxlen)
        row, val = o.hash_row(), o.hash_val()
        dense = o.dense_table() if name not in HASHED_ONLY else None

        def run(data):
            r = ob.Reference.match_hash(data, row, val, F, init, omp=True)
            if dense is not None:      # the two table layouts of the reference agree (time-driven == space-driven)
                assert np.array_equal(r, ob.Reference.match_dense(data, dense, F, init, omp=True))
            return r

        for s in range(SLICES[name]):
            t0 = time.time()
            data = np.empty(n + overlap, dtype=np.uint8)
            data[:n] = cfg.input_slice(n, s)
            data[n:] = cfg.input_slice(overlap, s + 1)
            inner = run(data)[:n]
            rec = {"workload": name, "slice": s, "n": n, "size_mib": args.size_mib, "overlap": overlap,
                   "patterns": F, "num_states": int(ref.t.num_states), "max_pattern_len": int(maxlen),
                   "pattern_file_fnv1a": wl.fnv1a(np.fromfile(pf, dtype=np.uint8)),
                   "input_fnv1a": wl.fnv1a(data[:n]), "successor_head_fnv1a": wl.fnv1a(data[n:]),
                   "inner": digest(inner, s * n)}
            # "last": the stream ends at n -- only walks that reach the end can differ
            w = min(n, 4 * overlap + 4096)
            tail = run(data[n - w:n].copy())
            last = inner.copy()
            last[n - w:] = tail
            keep = n - w + w // 2              # positions before this cannot reach the end: identical by construction
            assert np.array_equal(inner[n - w:keep], tail[: keep - (n - w)])
            rec["last"] = digest(last, s * n)
            rec["last_differs"] = bool(rec["last"]["fnv1a64"] != rec["inner"]["fnv1a64"])
            doc["records"] = [r for r in doc["records"]
                              if not (r["workload"] == name and r["slice"] == s and r["size_mib"] == args.size_mib)] + [rec]
            json.dump(doc, open(args.out, "w"), indent=1)
            print(name, "slice", s, "matches", rec["inner"]["match_count"], rec["last"]["match_count"],
                  "%.1fs" % (time.time() - t0), flush=True)
    print("wrote", args.out)


if __name__ == "__main__":
    main()

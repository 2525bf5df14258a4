#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors.json with the REFERENCE's own CPU code.

Runs only in the build container (needs oracle/_ref/libpfac_ref.so, i.e. /root/reference): the
reference's parsePatternFile + create_PFACTable_spaceDriven build the automaton, the reference's
PFAC_CPU_timeDriven / PFAC_CPU_OMP_spaceDriven produce the results.  Only data is written: seeds,
checksums of the generated inputs, and the sparse (position, pattern ID) results.

    python tests/golden/make_golden.py
"""

import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as ob              # noqa: E402
from pfac_amd import workloads as wl           # noqa: E402
from tests.test_oracle_golden import _golden_input   # noqa: E402

CASES = [
    {"name": "c2_random", "patterns": {"fn": "random_patterns", "args": [1000]},
     "input": {"kind": "random", "n": 262144 + 13, "seed": wl.SEED_C2_INPUT,
               "planted": [[1000 + 997 * i, i] for i in range(0, 240, 3)] + [[262144 + 13 - 8, 5]]}},
    {"name": "c3_http", "patterns": {"fn": "snort_patterns", "args": [2000]},
     "input": {"kind": "http", "n": 196608 + 1, "seed": wl.SEED_C3_INPUT, "pool": {"pool_size": 256, "embed_fraction": 0.25}}},
    {"name": "c5_adversarial", "patterns": {"fn": "adversarial_patterns", "args": [150]},
     "input": {"kind": "adversarial", "n": 65536, "seed": wl.SEED_C5, "pool": {"pool_size": 128}}},
]


def main():
    assert ob.have_reference(), "build oracle/_ref first: make -C oracle ref"
    tmp = tempfile.mkdtemp()
    out = {"_doc": "results of the reference's CPU matchers (oracle/_ref) on seeded inputs; see make_golden.py",
           "cases": []}
    for case in CASES:
        pats = getattr(wl, case["patterns"]["fn"])(*case["patterns"]["args"])
        pf = wl.write_pattern_file(os.path.join(tmp, case["name"] + ".pat"), pats)
        data = _golden_input(wl, case, pats)
        ref = ob.Reference(pf)
        o = ob.Oracle(pf)       # tables only: the reference's table builders need libcudart
        assert ref.edges() == o.edges()
        dense = ob.Reference.match_dense(data, o.dense_table(), ref.t.num_patterns, ref.t.initial_state, omp=False)
        hashed = ob.Reference.match_hash(data, o.hash_row(), o.hash_val(), ref.t.num_patterns, ref.t.initial_state, omp=True)
        assert np.array_equal(dense, hashed)
        pos = np.nonzero(dense)[0]
        fnv, cnt = ob.digest(dense)
        rec = dict(case)
        rec.update({
            "pattern_file_fnv1a": wl.fnv1a(np.fromfile(pf, dtype=np.uint8)),
            "input_fnv1a": wl.fnv1a(data),
            "num_states": int(ref.t.num_states), "max_pattern_len": int(ref.t.max_pattern_len),
            "positions": pos.tolist(), "ids": dense[pos].tolist(), "result_fnv1a": fnv,
        })
        assert cnt == pos.size
        out["cases"].append(rec)
        print(case["name"], "matches:", pos.size, "states:", ref.t.num_states)
    path = os.path.join(ROOT, "tests", "golden", "ref_vectors.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

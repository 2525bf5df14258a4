"""Pin the oracle (oracle/pfac_oracle.c) before anything is compared against it.

1. the reference's own known answers (tests/golden/known_answers.json cites each source):
   README.md:113-120, user guide r1.2 p.21 / p.27 / p.29, PFAC_hash_draft.pdf Fig. 1;
2. golden vectors produced by the reference's real CPU code (oracle/_ref, built from the unmodified
   reference sources) with tests/golden/make_golden.py;
3. when oracle/_ref is present: a live cross-check of the restatement against the reference's
   parser / trie builder / scalar and OpenMP matchers on the seeded workloads.
"""

import json
import os

import numpy as np
import pytest

from oracle import binding as ob


@pytest.fixture(scope="module")
def known(golden_dir):
    return json.load(open(os.path.join(golden_dir, "known_answers.json")))


def _load(golden_dir, ka):
    pf = os.path.join(golden_dir, ka["pattern_file"])
    data = np.fromfile(os.path.join(golden_dir, ka["input_file"]), dtype=np.uint8)
    return pf, data


def test_readme_example_all_four_cpu_variants(golden_dir, known):
    """README.md:113-120 + user guide p.27: ABEDEDABG -> {1,3,4,0,4,0,2,0,0}."""
    ka = known["example1"]
    pf, data = _load(golden_dir, ka)
    o = ob.Oracle(pf)
    for hashed in (False, True):
        for omp in (False, True):
            got = o.match(data, hashed=hashed, omp=omp)
            assert got.tolist() == ka["result_full"]
            assert got[:9].tolist() == ka["result_first9"]
    assert (o.num_states, o.initial_state, o.num_patterns) == (ka["num_states"], ka["initial_state"], ka["num_final"])


def test_reduce_known_answer(golden_dir, known):
    """user guide r1.2 p.29: h_num_matched = 5, h_pos = {0,1,2,4,6}, h_matched_result = {1,3,4,4,2}."""
    ka = known["example1"]
    pf, data = _load(golden_dir, ka)
    ids, pos = ob.reduce(ob.Oracle(pf).match(data))
    assert pos.tolist() == ka["reduce_pos"] and ids.tolist() == ka["reduce_id"]


def test_transition_table_dump_matches_user_guide_p21(golden_dir, tmp_path, known):
    """PFAC_dumpTransitionTable listing of user guide r1.2 p.21, byte for byte."""
    pf, _ = _load(golden_dir, known["example1"])
    out = tmp_path / "table.txt"
    ob.Oracle(pf).dump_table(str(out))
    want = open(os.path.join(golden_dir, "userguide_r1.2_p21_table.txt"), "rb").read()
    assert out.read_bytes() == want


def test_hash_paper_figure1_facts(golden_dir, known, tmp_path):
    """PFAC_hash_draft.pdf section I: 14 states (labelled 1..14), initial 11, (11,'h')->2, 'hershey'."""
    ka = known["example2"]
    pf, data = _load(golden_dir, ka)
    o = ob.Oracle(pf)
    assert (o.num_states, o.initial_state, o.num_patterns, o.num_leaves) == (
        ka["num_states"], ka["initial_state"], ka["num_final"], ka["num_leaves"])
    edges = set(o.edges())
    for s, ch, nx in ka["edges_required"]:
        assert (s, ord(ch), nx) in edges
    # survey-recorded output of the unmodified reference CPU sources
    for hashed in (False, True):
        assert o.match(data, hashed=hashed).tolist() == ka["result_full"]
    ids, pos = ob.reduce(o.match(data))
    assert [[int(p), int(i)] for p, i in zip(pos, ids)] == ka["reduce"]
    h = ka["hershey"]
    got = o.match(np.frombuffer(h["input"].encode(), dtype=np.uint8))
    patterns = open(pf, "rb").read().split(b"\n")
    assert patterns[got[0] - 1].decode() == h["pos0_pattern"] and got[1] == h["pos1"]


def test_golden_vectors_from_reference_build(golden_dir, workdir):
    """tests/golden/ref_vectors.json: seeded inputs and the reference's own sparse results."""
    from pfac_amd import workloads as wl
    path = os.path.join(golden_dir, "ref_vectors.json")
    vec = json.load(open(path))
    for case in vec["cases"]:
        pats = getattr(wl, case["patterns"]["fn"])(*case["patterns"]["args"])
        pf = wl.write_pattern_file(os.path.join(workdir, "golden_" + case["name"] + ".pat"), pats)
        assert wl.fnv1a(np.fromfile(pf, dtype=np.uint8)) == case["pattern_file_fnv1a"], "generator drifted"
        data = _golden_input(wl, case, pats)
        assert wl.fnv1a(data) == case["input_fnv1a"], "generator drifted"
        o = ob.Oracle(pf)
        assert (o.num_states, o.max_pattern_len) == (case["num_states"], case["max_pattern_len"])
        for hashed in (False, True):
            got = o.match(data, hashed=hashed, omp=True)
            ids, pos = ob.reduce(got)
            assert pos.tolist() == case["positions"], case["name"]
            assert ids.tolist() == case["ids"], case["name"]
            fnv, cnt = ob.digest(got)
            assert (fnv, cnt) == (case["result_fnv1a"], len(case["positions"]))
        o.close()


def _golden_input(wl, case, pats):
    kind = case["input"]["kind"]
    n, seed = case["input"]["n"], case["input"]["seed"]
    if kind == "random":
        data = wl.random_bytes(n, seed).copy()
    elif kind == "http":
        data = wl.http_stream(n, wl.http_message_pool(pats, **case["input"]["pool"]), seed)
    elif kind == "adversarial":
        data = wl.adversarial_stream(n, wl.adversarial_pool(pats, **case["input"]["pool"]), seed)
    else:
        raise ValueError(kind)
    for at, pid in case["input"].get("planted", []):
        p = np.frombuffer(pats[pid], dtype=np.uint8)[: data.size - at]   # may be cut by the end of the input
        data[at:at + p.size] = p
    return data


@pytest.mark.skipif(not ob.have_reference(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name", ["c1", "ex2", "c2", "c3", "c5", "dense_hits", "binary"])
def test_restatement_equals_reference_code(workloads, oracle_results, name):
    """Trie (state numbering, edge order) identical to the reference's parsePatternFile +
    create_PFACTable_spaceDriven; the reference's scalar and OpenMP matchers agree with the
    restatement on both table layouts."""
    w = workloads[name]
    o = ob.Oracle(w.pattern_file)
    r = ob.Reference(w.pattern_file)
    assert (r.t.num_patterns, r.t.num_states, r.t.initial_state, r.t.max_pattern_len) == (
        o.num_patterns, o.num_states, o.initial_state, o.max_pattern_len)
    assert r.edges() == o.edges()
    want = oracle_results[name]
    for omp in (False, True):
        assert np.array_equal(ob.Reference.match_dense(w.data, o.dense_table(), o.num_patterns, o.initial_state, omp), want)
        assert np.array_equal(ob.Reference.match_hash(w.data, o.hash_row(), o.hash_val(), o.num_patterns,
                                                      o.initial_state, omp), want)
    assert np.array_equal(o.match(w.data, hashed=True), want)
    assert np.array_equal(o.match(w.data, hashed=True, omp=True), want)


def test_oracle_rejects_what_the_reference_leaves_undefined(tmp_path):
    """Blank line before another pattern: reference asserts (PFAC_reorder_Table.cpp:291);
    duplicates: comparator is not a strict weak order (:63-64).  Trailing garbage without a newline
    is silently dropped (:181-195); trailing blank lines are harmless."""
    def load(b):
        p = tmp_path / "p.pat"
        p.write_bytes(b)
        return ob.Oracle(str(p))
    with pytest.raises(ob.OracleError) as e:
        load(b"AB\n\nCD\n")
    assert e.value.status == 10004
    with pytest.raises(ob.OracleError) as e:
        load(b"\nAB\n")
    assert e.value.status == 10004
    with pytest.raises(ob.OracleError) as e:
        load(b"AB\nCD\nAB\n")
    assert e.value.status == 10010
    assert load(b"AB\nCD\nEF").num_patterns == 2
    assert load(b"AB\nCD\n\n\n").num_patterns == 2
    assert load(b"").num_patterns == 0

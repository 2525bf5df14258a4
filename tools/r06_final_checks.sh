#!/bin/bash
# tools/r06_final_checks.sh [soak runs] -- what ran on the round's final build besides tools/measure_round.sh and tools/soak.sh (GPU box only): the suite under every forced
# walker, random call shapes against the reference-shaped kernel with the veto kernel forced (text and near misses), more seeds of the fuzzed pattern sets
O=gpurun_out/r06_final; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/suite_default.txt 2>&1; grep -E "passed|failed|FAILED" $O/suite_default.txt | tail -2
for w in veto stage window; do
  PFAC_TEST_WALKER=$w python -m pytest tests -m gpu -x -q > $O/suite_walker_$w.txt 2>&1; echo "PFAC_TEST_WALKER=$w: $(grep -E 'passed|failed' $O/suite_walker_$w.txt | tail -1)"; grep -E "^FAILED" $O/suite_walker_$w.txt
done
PFAC_TEST_WALKER=veto timeout 900 python tools/stress_calls.py 200 11 c6 2>&1 | tail -2
PFAC_TEST_WALKER=veto timeout 900 python tools/stress_calls.py 200 12 c3 2>&1 | tail -2
timeout 900 python tools/stress_calls.py 200 13 c6 2>&1 | tail -2
timeout 900 python tools/stress_fuzz.py 6000 40 2>&1 | tail -2
bash tools/soak.sh ${1:-8} $O/soak_final_build.txt | tail -3

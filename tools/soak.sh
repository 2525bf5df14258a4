#!/bin/bash
# tools/soak.sh [runs] [out]  -- the `-m gpu` suite N times on one lease (GPU box only): one line per run (outcome, seconds), the failures' names if any.
# What VERDICT round 5 asked for: a deterministic suite is shown by running it, not by saying so (profiles/r06_soak.txt).
N=${1:-30}; OUT=${2:-gpurun_out/soak.txt}; : > $OUT
for i in $(seq 1 $N); do
  t0=$(date +%s)
  python -m pytest tests -m gpu -q -p no:cacheprovider > /tmp/soak_run.txt 2>&1
  rc=$?
  echo "run $i rc $rc $(( $(date +%s) - t0 )) s: $(grep -E 'passed|failed' /tmp/soak_run.txt | tail -1)" >> $OUT
  grep -E "^FAILED|^ERROR" /tmp/soak_run.txt >> $OUT
done
echo "== $(grep -c ' rc 0 ' $OUT) of $N runs green" >> $OUT
cat $OUT

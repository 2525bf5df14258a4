#!/bin/bash
# first run of the tiled kernel: parity subset, hostile tests, naive-variant bench lines, small-input latency
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04a/parity.txt
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -s -k "snort_length or every_position or misaligned or duplicates" 2>&1 | tail -30 > gpurun_out/r04a/hostile.txt
for w in c3 c2 c5; do
  timeout 300 python bench.py --workload $w --variant naive --steps 5 --warmup 2 --pmc off --no-cpu-baseline --no-other-configs > gpurun_out/r04a/bench_naive_$w.json 2> gpurun_out/r04a/bench_naive_$w.err
done
timeout 300 python tools/small_input_latency.py > gpurun_out/r04a/latency.txt 2>&1
tail -5 gpurun_out/r04a/parity.txt; tail -12 gpurun_out/r04a/hostile.txt; cat gpurun_out/r04a/latency.txt | head -30
for w in c3 c2 c5; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04a/bench_naive_$w.json").read().strip().splitlines()[-1])
    print("$w", d["value"], d["ms_per_step"], d["config"].get("bit_exact"), d["config"].get("kernel_launched"))
except Exception as e:
    print("$w", "ERR", e, open("gpurun_out/r04a/bench_naive_$w.err").read()[-800:])
PY
done

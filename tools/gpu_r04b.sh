#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r04d/parity.txt
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -s -k "snort_length or every_position" 2>&1 | grep -E "GB/s|passed|failed|Error|assert" > gpurun_out/r04d/hostile.txt
for w in c3 c2 c5; do
  timeout 300 python bench.py --workload $w --variant naive --steps 5 --warmup 2 --pmc off --no-cpu-baseline --no-other-configs > gpurun_out/r04d/bench_naive_$w.json 2> gpurun_out/r04d/bench_naive_$w.err
done
for w in c3 c2 c5; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04d/bench_naive_$w.json").read().strip().splitlines()[-1])
    print("$w", d["value"], d["ms_per_step"], d["config"].get("bit_exact"), d["config"].get("kernel_launched"))
except Exception as e:
    print("$w", "ERR", e, open("gpurun_out/r04d/bench_naive_$w.err").read()[-800:])
PY
done
timeout 300 python tools/small_input_latency.py > gpurun_out/r04d/latency.txt 2>&1; grep -v filter gpurun_out/r04d/latency.txt

#!/bin/bash
O=gpurun_out/r02w; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
timeout 300 python tools/small_input_latency.py > $O/small_input_latency.txt 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?" >> $O/bench_default.err
tail -3 $O/pytest_gpu.txt; cat $O/small_input_latency.txt; tail -4 $O/bench_default.err
python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['placement_spread_kernel_ms'], d['host_path_pcie_inclusive'], d['reduce_api'])"

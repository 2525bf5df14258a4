#!/bin/bash
O=gpurun_out/r02host; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 900 python bench.py --no-cpu-baseline --pmc off > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['host_path_pcie_inclusive']); print(d['value'], d['config']['bit_exact'])"

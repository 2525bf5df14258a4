#!/bin/bash
# what the driver runs at round end: GPU suite, smoke(), default bench
O=gpurun_out/r02final; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -2 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'frac', round(r['frac'],3), 'traffic', r.get('traffic'), 'exact', d['config']['bit_exact'], 'stream', r.get('bare_stream_1r4w'), 'host', d.get('host_path_pcie_inclusive'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"

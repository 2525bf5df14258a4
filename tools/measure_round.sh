#!/bin/bash
# tools/measure_round.sh <tag>  -- the measurement set committed under profiles/ each round (GPU box only):
# bench lines, rocprofv3 kernel stats of the same command, PMC HBM traffic.  Everything under `timeout`.
T=${1:-r01}; O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err
timeout 300 python bench.py --workload c2 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --workload c5 --cpu-seconds 5 > $O/bench_c5_dense.json 2> $O/bench_c5_dense.err
timeout 300 python bench.py --workload c5 --perf-mode hash --cpu-seconds 5 > $O/bench_c5_hashed.json 2> $O/bench_c5_hashed.err
timeout 300 python bench.py --variant naive --no-cpu-baseline --steps 5 > $O/bench_c3_naive.json 2> /dev/null
timeout 300 python bench.py --workload c2 --variant naive --no-cpu-baseline --steps 5 > $O/bench_c2_naive.json 2> /dev/null
for w in c3 c2; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o prof -- python3 bench.py --worker rank --workload $w --no-other-configs > $O/bench_${w}_under_rocprof.json 2> $O/prof_$w.err
  timeout 400 tools/pmc_traffic.sh $w $O/traffic_$w > $O/traffic_$w.log 2>&1
done
grep -h '"metric"' $O/bench_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:40], '|', d['config']['kernel'], d['config']['table'], '|', d['value'], 'GB/s', r['kernel_ms_avg'], 'ms frac', round(r['frac'], 3), 'exact', d['config']['bit_exact'], 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'reduce', (d.get('reduce_api') or {}).get('value'))"

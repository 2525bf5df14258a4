#!/bin/bash
# tools/reduce_trace.sh [workload] -- kernel trace of the compacted-output call (PFAC_matchFromDeviceReduce) inside
# bench.py's rank worker: every launch of the last calls with its duration and the idle time in front of it.
W=${1:-c3}; O=gpurun_out/prof_reduce_$W; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o prof -- python3 bench.py --worker rank --workload $W --no-other-configs --steps 3 --warmup 1 --no-cpu-baseline --pmc off > $O.json 2> $O.err
python3 - "$O/prof_kernel_trace.csv" <<PY
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "pfac_order" in r["Kernel_Name"] or "radix" in r["Kernel_Name"])
prev = None
for r in rows[max(0, last - 26):last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    name = name[name.find("pfac"):][:44] if "pfac" in name else name[:44]
    print(f"{name:46s} {(e - s) / 1e3:9.2f} us   idle before {0 if prev is None else (s - prev) / 1e3:8.2f} us")
    prev = e
PY
python3 -c "
import json
d = json.loads([l for l in open('$O.json') if l.startswith('{')][-1]); r = d['reduce_api']
print({k: r[k] for k in ('ms_per_call', 'gpu_ms', 'host_overhead_ms', 'matches', 'same_result_as_full_vector')})"

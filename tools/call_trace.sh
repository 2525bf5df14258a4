#!/bin/bash
# tools/call_trace.sh <MiB> [auto|filter|naive] -- kernel trace of back-to-back PFAC_matchFromDevice calls of one size: every launch of the last
# calls with its duration and the idle time in front of it (what a call costs beyond its kernels).
M=${1:-64}; V=${2:-auto}; O=gpurun_out/prof_call_${M}_$V; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o prof -- python3 tools/call_trace_driver.py $M $V 12 > $O.out 2> $O.err
python3 - "$O/prof_kernel_trace.csv" <<PY
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "pfac" in r["Kernel_Name"]]
prev = None
for r in rows[-8:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]; name = name[name.find("pfac"):][:60]
    print(f"{name:62s} {(e - s) / 1e3:9.2f} us   idle before {0 if prev is None else (s - prev) / 1e3:8.2f} us")
    prev = e
PY

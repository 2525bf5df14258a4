#!/bin/bash
# tools/pmc_sets.sh <workload> <outdir> "<set1 counters>" "<set2 counters>" ...  (GPU box only)
W=$1; OUT=$2; shift 2; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -o s$i -- python3 bench.py --worker pmc --workload $W --no-verify > $OUT/s$i.json 2> $OUT/s$i.err
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pfac_scan_filter" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("%-36s %.4g" % (k, sum(v) / len(v)))
PY

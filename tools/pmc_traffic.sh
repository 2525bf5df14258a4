#!/bin/bash
# tools/pmc_traffic.sh <workload> <outdir> -- HBM traffic of the scan kernel from rocprofv3 PMC counters
# (GPU box only).  FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots: 3 + 2).  The
# stream probe is profiled in the same way to calibrate the counters on known byte counts
# (MI355X_MICROARCH.md "HBM": FETCH_SIZE under-reports wide coalesced reads 2x on gfx950).
W=$1; OUT=$2; mkdir -p $OUT /tmp/pb
export TMPDIR=/tmp PATH=/opt/rocm/bin:$PATH
hipcc -O3 --offload-arch=gfx950 -o /tmp/pb/stream_probe tools/stream_probe.hip > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/scan_$c -o scan -- python3 bench.py --worker pmc --workload $W --no-verify > $OUT/scan_$c.json 2> $OUT/scan_$c.err
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/probe_$c -o probe -- /tmp/pb/stream_probe > $OUT/probe_$c.log 2> $OUT/probe_$c.err
done
python3 - "$OUT" "$W" <<'PY'
import csv, glob, collections, json, sys
out, w = sys.argv[1], sys.argv[2]
def mean_by_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    # bench.py --worker pmc: one synchronised launch (it votes on the handle's walker), then 4; an instance launched once is that first launch
    many = [k for k, v in agg.items() if "pfac_scan_filter" in k and len(v) > 1]
    return {k: sum(v) / len(v) for k, v in agg.items() if not ("pfac_scan_filter" in k and many and k not in many)}
res = {"workload": w, "unit": "counter KB (1 KB = 1024 B) per launch"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    scan = mean_by_kernel(glob.glob(f"{out}/scan_{c}/*counter_collection.csv")[0])
    probe = mean_by_kernel(glob.glob(f"{out}/probe_{c}/*counter_collection.csv")[0])
    res[c] = {"scan": {k.split("(")[0] + ("(" + k.split("(")[1] if k.startswith("void (") else ""): v for k, v in scan.items() if "pfac_scan" in k},
              "probe": {k[:60]: v for k, v in probe.items()}}
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY

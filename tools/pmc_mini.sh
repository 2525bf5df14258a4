#!/bin/bash
export TMPDIR=/tmp
for so in "$@"; do
  name=$(basename $so .so); cp $so pfac_amd/lib/libpfac_gfx950.so
  OUT=gpurun_out/pmc_mini/$name; mkdir -p $OUT
  timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/s1 -o s1 -- python3 bench.py --worker pmc --workload c3 --no-verify > $OUT/s1.json 2> $OUT/s1.err
done
python3 - "$@" <<'PY'
import csv, glob, collections, sys, os
names = [os.path.basename(p)[:-3] for p in sys.argv[1:]]
table = collections.OrderedDict()
for n in names:
    for f in sorted(glob.glob("gpurun_out/pmc_mini/%s/*/*_counter_collection.csv" % n)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pfac_scan_filter" in r["Kernel_Name"] and "true, 4" not in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            table.setdefault(k, {})[n] = sum(v) / len(v)
print("%-22s" % "counter" + "".join("%14s" % n for n in names))
for k, row in table.items():
    print("%-22s" % k + "".join("%14.4g" % row.get(n, float('nan')) for n in names))
PY

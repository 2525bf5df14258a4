"""tools/placement_pmc_report.py <rocprofv3 output dir> ...  -- per (input, result) pair: mean kernel time and counters"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(trace)):
        if "pfac_scan_filter" in r["Kernel_Name"]: dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    ids = sorted(dur, key=int)[10:]                      # skip the 10 settle launches
    val = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc)):
        if r["Dispatch_Id"] in dur: val[r["Counter_Name"]][r["Dispatch_Id"]] = float(r["Counter_Value"])
    names = sorted(val)
    print(d); print("pair      ms   " + " ".join("%14s" % x[:14] for x in names))
    for p in range(8):
        sel = ids[p * 6 + 1:p * 6 + 6]
        if not sel: continue
        print("in%d/out%d %.3f " % (p // 4, p % 4, sum(dur[i] for i in sel) / len(sel)) + " ".join("%14.4g" % (sum(val[x].get(i, 0) for i in sel) / len(sel)) for x in names))

#!/usr/bin/env python3
"""tools/pmc_variants.py [--workload c3] [--counters A,B,...] variant ...   (GPU box only)

One rocprofv3 --pmc pass (+ kernel trace) per variant of tools/bin/variants/ ("tree" = pfac_amd/lib as it is) over
`bench.py --worker pmc` (4 launches of the full-result scan kernel); prints the per-launch mean of every counter, the
kernel duration from the trace and the effective shader clock GRBM_GUI_ACTIVE / duration (MI355X_MICROARCH.md, "DVFS
give-back").  The table also goes to gpurun_out/pmc_<tag>.txt."""
import argparse, collections, csv, glob, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pfac_amd", "lib")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--tag", default="pmc")
    ap.add_argument("--counters", default="GRBM_GUI_ACTIVE,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD")
    ap.add_argument("specs", nargs="+")
    a = ap.parse_args()
    keep = os.path.join(ROOT, "tools", "bin", "variants", "_tree")
    os.makedirs(keep, exist_ok=True)
    for f in ("libpfac.so", "libpfac_gfx950.so"):
        shutil.copy2(os.path.join(LIB, f), os.path.join(keep, f))
    table = collections.OrderedDict()
    env0 = dict(os.environ, TMPDIR="/tmp")
    try:
        for spec in a.specs:
            name, *envs = spec.split(",")
            src = keep if name == "tree" else os.path.join(ROOT, "tools", "bin", "variants", name)
            for f in ("libpfac.so", "libpfac_gfx950.so"):
                if os.path.exists(os.path.join(src, f)):
                    shutil.copy2(os.path.join(src, f), os.path.join(LIB, f))
            env = dict(env0)
            env.update(e.split("=", 1) for e in envs)
            out = os.path.join(ROOT, "gpurun_out", "pmc_variants", a.tag, name)
            shutil.rmtree(out, ignore_errors=True)
            cmd = ["rocprofv3", "--pmc"] + a.counters.split(",") + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--worker", "pmc", "--workload", a.workload, "--no-verify"]
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=90)
            except subprocess.TimeoutExpired:
                table[spec] = {"FAILED": 1}
                continue
            # bench.py --worker pmc: one synchronised launch (it votes on the walker of the next ones), then 4 -- the instance launched most
            names = collections.Counter()
            for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "pfac_scan_filter" in r["Kernel_Name"]:
                        names[r["Kernel_Name"]] += 1
            mine = names.most_common(1)[0][0] if names else "pfac_scan_filter"
            agg = collections.defaultdict(list)
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Kernel_Name"] == mine:
                        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur = []
            for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Kernel_Name"] == mine:
                        dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            row = {k: sum(v) / len(v) for k, v in agg.items()}
            if dur:
                row["duration_us"] = sum(dur) / len(dur) / 1e3
                if "GRBM_GUI_ACTIVE" in row:
                    row["clock_GHz"] = row["GRBM_GUI_ACTIVE"] / (sum(dur) / len(dur))
            table[spec] = row
    finally:
        for f in ("libpfac.so", "libpfac_gfx950.so"):
            shutil.copy2(os.path.join(keep, f), os.path.join(LIB, f))
    keys = []
    for row in table.values():
        for k in row:
            if k not in keys:
                keys.append(k)
    lines = ["%-22s" % "counter" + "".join("%16s" % s[:15] for s in table)]
    for k in keys:
        lines.append("%-22s" % k + "".join("%16.5g" % table[s].get(k, float("nan")) for s in table))
    text = "\n".join(lines)
    print(text)
    open(os.path.join(ROOT, "gpurun_out", "pmc_%s.txt" % a.tag), "w").write(text + "\n")


if __name__ == "__main__":
    main()

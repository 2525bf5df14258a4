#!/usr/bin/env python3
"""tools/pmc_run.py --kernel SUBSTR [--counters A,B,..] [--tag T] -- script.py args...   (GPU box only)
rocprofv3 --pmc passes (at most 8 counters per pass; the passes are separate runs) + one --kernel-trace run over
`python3 script.py args`; prints the per-launch mean of every counter for kernels whose name contains SUBSTR, the mean
duration and the effective clock.  No --pmc together with any trace other than --kernel-trace (the pool refuses it)."""
import argparse, collections, csv, glob, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = "GRBM_GUI_ACTIVE,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD;SQ_INSTS_VMEM_WR,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_WAIT_INST_ANY,SQ_INST_CYCLES_VMEM,SQ_INSTS_SMEM"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--counters", default=DEFAULT, help="passes separated by ';', counters by ','")
    ap.add_argument("--tag", default="pmc_run")
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    rest = a.rest[1:] if a.rest and a.rest[0] == "--" else a.rest
    if rest and not os.path.isabs(rest[0]):
        rest[0] = os.path.join(ROOT, rest[0])
    env = dict(os.environ, TMPDIR="/tmp")
    row = collections.OrderedDict()
    passes = [p for p in a.counters.split(";") if p]
    for i, p in enumerate(passes + [None]):
        out = os.path.join(ROOT, "gpurun_out", "pmc_run", a.tag, "pass%d" % i)
        shutil.rmtree(out, ignore_errors=True)
        cmd = ["rocprofv3"] + (["--pmc"] + p.split(",") if p else ["--kernel-trace"]) + ["--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable] + rest
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
        except subprocess.TimeoutExpired:
            row["pass%d_TIMEOUT" % i] = 1
            continue
        agg = collections.defaultdict(list)
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if a.kernel in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            row[k] = sum(v) / len(v)
        dur = []
        for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if a.kernel in r["Kernel_Name"]:
                    dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        if dur:
            row["launches"] = len(dur)
            row["duration_us"] = sum(dur) / len(dur) / 1e3
            if "GRBM_GUI_ACTIVE" in row:
                row["clock_GHz"] = row["GRBM_GUI_ACTIVE"] / (sum(dur) / len(dur))
    text = "\n".join("%-24s %16.6g" % (k, v) for k, v in row.items())
    print("# " + " ".join(rest) + "  kernel~" + a.kernel)
    print(text)
    open(os.path.join(ROOT, "gpurun_out", "pmc_run_%s.txt" % a.tag), "w").write("# " + " ".join(rest) + "\n" + text + "\n")


if __name__ == "__main__":
    main()

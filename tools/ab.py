#!/usr/bin/env python3
"""tools/ab.py [--workloads c3,c2] [--repeat 3] [--steps 20] [--tag NAME] spec ...   (GPU box only)

Bench pre-built variants against each other.  spec = name[,ENV=VALUE,...]: tools/bin/variants/<name>/ (built in the
container by tools/build_variant.sh) holds libpfac_gfx950.so and libpfac.so, which are swapped into pfac_amd/lib for the
run; the ENV settings (e.g. the PFAC_DBG_* knobs of the pattern compiler) are exported to the bench process.  The name
"tree" stands for the libraries already in pfac_amd/lib.  Runs are interleaved (variant A, B, A, B, ...), REPEAT
processes each: run-to-run spread of one build is a few per cent on this pool, single runs cannot rank variants.
Prints min / median kernel ms, exactness, walks started, and the spread over the four buffer pairs; the table also goes
to gpurun_out/ab_<tag>.txt."""
import argparse, json, os, shutil, statistics, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pfac_amd", "lib")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c3,c2")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--tag", default="ab")
    ap.add_argument("--run-timeout", type=int, default=120, help="seconds per bench process")
    ap.add_argument("--extra", default="", help="extra bench.py arguments")
    ap.add_argument("specs", nargs="+")
    a = ap.parse_args()
    keep = os.path.join(ROOT, "tools", "bin", "variants", "_tree")
    os.makedirs(keep, exist_ok=True)
    for f in ("libpfac.so", "libpfac_gfx950.so"):
        shutil.copy2(os.path.join(LIB, f), os.path.join(keep, f))
    rows = {}
    dead = set()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    raw = open(os.path.join(ROOT, "gpurun_out", "ab_%s_raw.txt" % a.tag), "w")
    try:
        for r in range(a.repeat):
            for spec in a.specs:
                name, *envs = spec.split(",")
                src = keep if name == "tree" else os.path.join(ROOT, "tools", "bin", "variants", name)
                for f in ("libpfac.so", "libpfac_gfx950.so"):
                    if os.path.exists(os.path.join(src, f)):
                        shutil.copy2(os.path.join(src, f), os.path.join(LIB, f))
                env = dict(os.environ)
                if name != "tree":
                    env["PFAC_AB_OLD_LIBS"] = "1"
                env.update(e.split("=", 1) for e in envs)
                for w in a.workloads.split(","):
                    ww, extra = (("c5", ["--perf-mode", "hash"]) if w == "c5h" else (w, []))
                    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "3", "--workload", ww,
                           "--no-cpu-baseline", "--no-other-configs", "--spread", "--pmc", "off"] + extra + a.extra.split()
                    if (spec, w) in dead:
                        continue
                    try:                                    # a variant that hangs costs one short timeout, once
                        p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=a.run_timeout)
                        d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
                    except Exception as e:
                        rows.setdefault((spec, w), []).append(None)
                        dead.add((spec, w))
                        raw.write("%s %s FAILED %s\n" % (spec, w, type(e).__name__)); raw.flush()
                        continue
                    rf = d["roofline"]
                    raw.write("%s %s %s exact %s reduce %s\n" % (spec, w, rf["kernel_ms_avg"], d["config"]["bit_exact"], (d.get("reduce_api") or {}).get("ms_per_call"))); raw.flush()
                    sp = list((rf.get("placement_spread_kernel_ms") or {"x": rf["kernel_ms_avg"]}).values())
                    ws = d["config"].get("walk_stats") or {}
                    rows.setdefault((spec, w), []).append((rf["kernel_ms_avg"], d["config"]["bit_exact"], (d.get("reduce_api") or {}).get("ms_per_call"),
                                                           min(sp), max(sp), ws.get("walksStarted"), (rf.get("bare_stream_1r4w") or {}).get("ms"), ws.get("walkerRounds"), ws.get("laneSteps"), ws.get("ladderCandidates"),
                                                           (d.get("reduce_api") or {}).get("kernel_ms")))
    finally:
        for f in ("libpfac.so", "libpfac_gfx950.so"):
            shutil.copy2(os.path.join(keep, f), os.path.join(LIB, f))
    lines = []
    for (spec, w), v in rows.items():
        ok = [x for x in v if x]
        if not ok:
            lines.append("%-44s %-4s FAILED (%d runs)" % (spec, w, len(v)))
            continue
        ms = [x[0] for x in ok]
        red = [x[2] for x in ok if x[2] is not None]
        stream = [x[6] for x in ok if x[6]]
        redk = [x[10] for x in ok if x[10] is not None]
        lines.append("%-44s %-4s kernel ms min %.4f median %.4f max %.4f (n=%d%s) exact %s | reduce call ms min %s kernel ms min/median %s | 4 buffer pairs %.4f .. %.4f | walks %s rounds %s lane steps %s candidates %s | bare stream %s" % (
            spec, w, min(ms), statistics.median(ms), max(ms), len(ok), ", %d failed" % (len(v) - len(ok)) if len(ok) < len(v) else "",
            all(x[1] for x in ok), ("%.3f" % min(red)) if red else "-", ("%.4f/%.4f" % (min(redk), statistics.median(redk))) if redk else "-", min(x[3] for x in ok), max(x[4] for x in ok), ok[-1][5], ok[-1][7], ok[-1][8], ok[-1][9],
            ("%.4f" % min(stream)) if stream else "-"))
    text = "\n".join(lines)
    print(text)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "ab_%s.txt" % a.tag), "w").write(text + "\n")


if __name__ == "__main__":
    main()

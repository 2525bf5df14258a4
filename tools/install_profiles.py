#!/usr/bin/env python3
"""tools/install_profiles.py <gpurun_out subdir> <round tag>  -- copy what tools/measure_round.sh
collected on the GPU box into profiles/ under the names DESIGN.md and bench.py refer to."""
import csv, json, os, shutil, sys
src, tag = os.path.join("gpurun_out", sys.argv[1]), sys.argv[2]
def cp(a, b):
    shutil.copy(os.path.join(src, a), os.path.join("profiles", f"{tag}_{b}"))
for a, b in (("bench_c3_default.json", "bench_c3_default.json"), ("bench_c2.json", "bench_c2.json"),
             ("bench_c5_dense.json", "bench_c5_dense.json"), ("bench_c5_hashed.json", "bench_c5_hashed.json"),
             ("bench_c6.json", "bench_c6.json"), ("bench_c6_walker_window.json", "bench_c6_walker_window.json"), ("bench_c6_walker_stage.json", "bench_c6_walker_stage.json"),
             ("bench_c6_walker_veto.json", "bench_c6_walker_veto.json"), ("bench_c6_under_rocprof.json", "bench_c6_under_rocprof.json"), ("pmc_full_c6.txt", "pmc_full_result_kernel_c6.txt"),
             ("bench_c3_naive.json", "bench_c3_naive_kernel.json"), ("bench_c2_naive.json", "bench_c2_naive_kernel.json"),
             ("bench_c3_under_rocprof.json", "bench_c3_under_rocprof.json"), ("bench_c2_under_rocprof.json", "bench_c2_under_rocprof.json"),
             ("bench_c3_2ranks_one_gpu_gloo.json", "bench_c3_2ranks_one_gpu_gloo.json"),
             ("bench_c5_naive.json", "bench_c5_naive_kernel.json"), ("bench_c4_strong_one_gpu.json", "bench_c4_strong_one_gpu.json"),
             ("bench_c5_under_rocprof.json", "bench_c5_under_rocprof.json"),
             ("bench_c5_walker_window.json", "bench_c5_walker_window.json"), ("bench_c5_walker_stage.json", "bench_c5_walker_stage.json"),
             ("bench_c3_walker_window.json", "bench_c3_walker_window.json"), ("bench_c3_walker_stage.json", "bench_c3_walker_stage.json"),
             ("bench_c2_reftable_dense.json", "bench_c2_reftable_dense.json"), ("bench_c3_reftable_hashed.json", "bench_c3_reftable_hashed.json"),
             ("pmc_full_c5.txt", "pmc_full_result_kernel_c5.txt"), ("pmc_full_c3.txt", "pmc_full_result_kernel_c3.txt"),
             ("host_numa_probe.txt", "host_numa_probe.txt"), ("small_input_latency.txt", "small_input_latency.txt"),
             ("reduce_ablation.txt", "reduce_kernel_ablation_pmc.txt"), ("level1_floor.txt", "level1_floor.txt")):
    if os.path.exists(os.path.join(src, a)):
        cp(a, b)
with open(os.path.join("profiles", f"{tag}_pmc_instruction_counts.txt"), "w") as out:
    out.write("# rocprofv3 --pmc (tools/pmc_run.py over tools/reduce_driver.py: 4 calls of PFAC_matchFromDeviceReduce on the 1 GiB stream), per-launch means of the\n"
              "# compacted-output scan kernel pfac_scan_filter<..., REDUCE = true, 2>; counters are summed over the 8 XCDs (GRBM_GUI_ACTIVE / 8 = cycles)\n")
    for w in ("c3", "c5"):
        f = os.path.join(src, f"pmc_reduce_{w}.txt")
        if os.path.exists(f):
            out.write(f"== {w}\n" + open(f).read() + "\n")
for w in ("c3", "c2", "c5", "c6"):
    if not os.path.exists(os.path.join(src, f"prof_{w}", "prof_kernel_stats.csv")):
        continue
    rows = list(csv.reader(open(os.path.join(src, f"prof_{w}", "prof_kernel_stats.csv"))))
    keep = [rows[0]] + [r for r in rows[1:] if "pfac_scan" in r[0] or "pfac_order" in r[0] or "fillBuffer" in r[0]]
    csv.writer(open(os.path.join("profiles", f"{tag}_{w}_rocprofv3_kernel_stats.csv"), "w")).writerows(keep)
    cp(os.path.join(f"traffic_{w}", "summary.json"), f"hbm_traffic_{w}.json")
    d = json.load(open(os.path.join("profiles", f"{tag}_hbm_traffic_{w}.json")))
    def full_result(k):   # pfac_scan_filter<TEX, HAS_SHORT, REDUCE, WALKS>: REDUCE = false
        targs = [a.strip() for a in k.split("<", 1)[-1].split(">", 1)[0].split(",")]
        return "pfac_scan_filter" in k and len(targs) >= 3 and targs[2] == "false"
    f = [v for k, v in d["FETCH_SIZE"]["scan"].items() if full_result(k)][0]
    wr = [v for k, v in d["WRITE_SIZE"]["scan"].items() if full_result(k)][0]
    print(w, "filter kernel rocprof avg ns:", [(r[0][28:64], r[1], r[3]) for r in keep[1:] if "pfac_scan_filter" in r[0]],
          "| traffic GB: 2xFETCH %.3f + WRITE %.3f = %.3f" % (2 * f * 1024 / 1e9, wr * 1024 / 1e9, (2 * f + wr) * 1024 / 1e9))

#!/bin/bash
# tools/measure_round.sh <tag>  -- the measurement set committed under profiles/ each round (GPU box only):
# bench lines (each with its own PMC traffic passes and CPU baseline), rocprofv3 kernel stats of the same
# workload run as a single rank, the 2-rank dry run.  Everything under `timeout`.
T=${1:-r06}; O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err
timeout 400 python bench.py --workload c2 --no-other-configs > $O/bench_c2.json 2> $O/bench_c2.err
timeout 400 python bench.py --workload c5 --no-other-configs > $O/bench_c5_dense.json 2> $O/bench_c5_dense.err
timeout 400 python bench.py --workload c5 --perf-mode hash --no-other-configs > $O/bench_c5_hashed.json 2> $O/bench_c5_hashed.err
# round 6: the near-miss stream over the Snort-scale set + the shared-prefix patterns (tail table in device memory: the VETO = 2 kernel under PFACX_WALKER_AUTO)
timeout 400 python bench.py --workload c6 --no-other-configs > $O/bench_c6.json 2> $O/bench_c6.err
for wk in window stage veto; do
  timeout 300 python bench.py --workload c6 --walker $wk --no-cpu-baseline --no-other-configs --pmc off > $O/bench_c6_walker_$wk.json 2> /dev/null
done
timeout 300 python bench.py --variant naive --no-cpu-baseline --no-other-configs --pmc off --steps 5 > $O/bench_c3_naive.json 2> /dev/null
timeout 300 python bench.py --workload c2 --variant naive --no-cpu-baseline --no-other-configs --pmc off --steps 5 > $O/bench_c2_naive.json 2> /dev/null
timeout 400 python bench.py --gpus 2 --dist-backend gloo --no-other-configs --pmc off > $O/bench_c3_2ranks_one_gpu_gloo.json 2> $O/bench_2ranks.err
timeout 300 python bench.py --workload c5 --variant naive --no-cpu-baseline --no-other-configs --pmc off --steps 5 > $O/bench_c5_naive.json 2> /dev/null
# round 5: the two walkers of the full-result kernel on the near-miss stream and on text (AUTO picks stage / window); the reference-layout tables behind the tiled frame
for wk in window stage; do
  timeout 300 python bench.py --workload c5 --walker $wk --no-cpu-baseline --no-other-configs --pmc off > $O/bench_c5_walker_$wk.json 2> /dev/null
  timeout 300 python bench.py --workload c3 --walker $wk --no-cpu-baseline --no-other-configs --pmc off > $O/bench_c3_walker_$wk.json 2> /dev/null
done
timeout 300 python bench.py --workload c2 --variant reftable --perf-mode dense --no-cpu-baseline --no-other-configs --pmc off --steps 5 > $O/bench_c2_reftable_dense.json 2> /dev/null
timeout 300 python bench.py --workload c3 --variant reftable --perf-mode hash --no-cpu-baseline --no-other-configs --pmc off --steps 5 > $O/bench_c3_reftable_hashed.json 2> /dev/null
timeout 300 python tools/pmc_variants.py --workload c5 --tag ${T}_c5_full tree > $O/pmc_full_c5.txt 2>&1
timeout 300 python tools/pmc_variants.py --workload c3 --tag ${T}_c3_full tree > $O/pmc_full_c3.txt 2>&1
timeout 300 python tools/pmc_variants.py --workload c6 --tag ${T}_c6_full tree > $O/pmc_full_c6.txt 2>&1
timeout 200 python tools/host_numa_probe.py > $O/host_numa_probe.txt 2>&1
timeout 600 python bench.py --scaling strong --total-gib 8 --steps 5 --warmup 2 > $O/bench_c4_strong_one_gpu.json 2> $O/bench_c4_strong.err
timeout 300 python tools/small_input_latency.py > $O/small_input_latency.txt 2>&1
for w in c3 c5; do
  python3 tools/pmc_run.py --kernel pfac_scan_filter --tag reduce_$w --counters "GRBM_GUI_ACTIVE,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_LDS_IDX_ACTIVE,SQ_LDS_BANK_CONFLICT,SQ_WAVE_CYCLES" -- tools/reduce_driver.py $w 4 > $O/pmc_reduce_$w.txt 2>&1
done
# round 6: the compacted-output kernel's ablation ladder (tools/build_variant.sh abl1 / abl2 WORK -DPFAC_ABLATE=1 / 2 in the container first) and the floor of its level 1
[ -d tools/bin/variants/abl1 ] && timeout 600 bash tools/gpu_pmc_variants_reduce.sh c3 tree abl1 abl2 > $O/reduce_ablation.txt 2>&1
hipcc -O3 --offload-arch=gfx950 -o /tmp/level1_floor tools/level1_floor.hip > /dev/null 2>&1 && timeout 300 /tmp/level1_floor > $O/level1_floor.txt 2>&1
timeout 400 tools/pmc_traffic.sh c5 $O/traffic_c5 > $O/traffic_c5.log 2>&1
for w in c3 c2 c5 c6; do
  # 500 timed launches: the ~35 launches in front of them (first launch, settling, warm-up) run at rising clocks, up to 25 % slower, and are in the profiler's average too
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o prof -- python3 bench.py --worker rank --workload $w --no-other-configs --steps 500 > $O/bench_${w}_under_rocprof.json 2> $O/prof_$w.err
  timeout 400 tools/pmc_traffic.sh $w $O/traffic_$w > $O/traffic_$w.log 2>&1
done
grep -h '"metric"' $O/bench_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:40], '|', d['n_gpus'], d['config']['kernel'], d['config']['table'], '|', d['value'], 'GB/s', r['kernel_ms_avg'], 'ms frac', round(r['frac'], 3), 'traffic', r.get('traffic'), 'exact', d['config']['bit_exact'], 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'reduce', (d.get('reduce_api') or {}).get('value'))"

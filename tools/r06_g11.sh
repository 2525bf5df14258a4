bash tools/timing_run.sh timing c6 c3 > gpurun_out/r06_timing_v6.txt 2>&1
cat gpurun_out/r06_timing_v6.txt
python tools/pmc_variants.py --workload c6 --tag r06b_c6 tree
python tools/pmc_variants.py --workload c5 --tag r06b_c5 tree

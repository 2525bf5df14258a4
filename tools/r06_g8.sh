set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests_6_full.txt 2>&1
grep -E "GB/s|passed|failed|FAILED|Error" gpurun_out/r06_gputests_6_full.txt | cut -c1-900
bash tools/timing_run.sh timing c6 c5 c3 > gpurun_out/r06_timing_v5.txt 2>&1
cat gpurun_out/r06_timing_v5.txt
for w in c5 c6 c3; do python tools/pmc_variants.py --workload $w --tag r06_$w tree; done

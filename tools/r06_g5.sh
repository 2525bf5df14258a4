set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gputests_3.txt
tail -n 4 gpurun_out/r06_gputests_3.txt
for w in c6 c5 c3 c2; do
python bench.py --workload $w --no-other-configs --no-cpu-baseline --steps 10 > gpurun_out/r06_${w}_v3_auto.json 2> gpurun_out/r06_${w}_v3_auto.log
done
python bench.py --workload c5 --perf-mode hash --no-other-configs --no-cpu-baseline --steps 10 > gpurun_out/r06_c5h_v3_auto.json 2> gpurun_out/r06_c5h_v3_auto.log
bash tools/timing_run.sh timing c6 c5 c3 > gpurun_out/r06_timing_v3.txt 2>&1
cat gpurun_out/r06_timing_v3.txt

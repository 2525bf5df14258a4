set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gputests_0.txt
for w in auto window stage; do
python bench.py --workload c6 --walker $w --no-other-configs --no-cpu-baseline --steps 10 > gpurun_out/r06_c6_base_$w.json 2> gpurun_out/r06_c6_base_$w.log
done
tail -3 gpurun_out/r06_gputests_0.txt

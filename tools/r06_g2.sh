set -x
for w in auto veto; do
python bench.py --workload c6 --walker $w --no-other-configs --no-cpu-baseline --steps 10 > gpurun_out/r06_c6_v1_$w.json 2> gpurun_out/r06_c6_v1_$w.log
done
python bench.py --workload c3 --walker veto --no-other-configs --no-cpu-baseline --steps 10 > gpurun_out/r06_c3_v1_veto.json 2> gpurun_out/r06_c3_v1_veto.log
python bench.py --workload c3 --no-other-configs --no-cpu-baseline --steps 10 > gpurun_out/r06_c3_v1_auto.json 2> gpurun_out/r06_c3_v1_auto.log
PFAC_TEST_WALKER=veto python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gputests_veto.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gputests_1.txt
tail -3 gpurun_out/r06_gputests_veto.txt gpurun_out/r06_gputests_1.txt

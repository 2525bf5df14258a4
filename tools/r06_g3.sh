set -x
for w in veto auto; do
python bench.py --workload c6 --walker $w --no-other-configs --no-cpu-baseline --steps 10 > gpurun_out/r06_c6_v2_$w.json 2> gpurun_out/r06_c6_v2_$w.log
done
python bench.py --workload c3 --walker veto --no-other-configs --no-cpu-baseline --steps 10 > gpurun_out/r06_c3_v2_veto.json 2> gpurun_out/r06_c3_v2_veto.log
python bench.py --workload c3 --no-other-configs --no-cpu-baseline --steps 10 > gpurun_out/r06_c3_v2_auto.json 2> gpurun_out/r06_c3_v2_auto.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gputests_2.txt
PFAC_TEST_WALKER=veto python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gputests_veto2.txt
tail -n 3 gpurun_out/r06_gputests_2.txt gpurun_out/r06_gputests_veto2.txt

set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gputests_4.txt
tail -n 4 gpurun_out/r06_gputests_4.txt
python tools/ab.py --workloads c3,c5,c6,c2 --repeat 2 --steps 10 --tag gate tree gate0 gate48 > gpurun_out/r06_ab_gate.log 2>&1
cat gpurun_out/ab_gate.txt
bash tools/timing_run.sh timing c6 c5 c3 > gpurun_out/r06_timing_v4.txt 2>&1
cat gpurun_out/r06_timing_v4.txt
python tools/ab.py --workloads c3 --repeat 2 --steps 5 --tag reduce1 tree ws1 rb32 rm32 ws1rb32 > gpurun_out/r06_ab_reduce1.log 2>&1
cat gpurun_out/ab_reduce1.txt

set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests_5_full.txt 2>&1
grep -E "GB/s|passed|failed|FAILED|Error" gpurun_out/r06_gputests_5_full.txt | cut -c1-900
python tools/ab.py --workloads c5,c6,c3 --repeat 2 --steps 10 --tag gate2 tree gate64t6 gate64t10 gate0 > gpurun_out/r06_ab_gate2.log 2>&1
cat gpurun_out/ab_gate2.txt
python bench.py --variant naive --workload c3 --no-other-configs --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 tiled alone', d['value'], d['roofline']['kernel_ms_avg'], d['config']['bit_exact'])"
python bench.py --variant naive --workload c5 --no-other-configs --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 tiled alone', d['value'], d['roofline']['kernel_ms_avg'], d['config']['bit_exact'])"

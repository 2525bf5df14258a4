set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests_7_full.txt 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r06_gputests_7_full.txt | cut -c1-400
python tools/ab.py --workloads c5,c6,c3,c2 --repeat 3 --steps 10 --tag blk tree noblk > gpurun_out/r06_ab_blk.log 2>&1
cat gpurun_out/ab_blk.txt

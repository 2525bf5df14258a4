bash tools/timing_run.sh timing c6 c5 c3 > gpurun_out/r06_timing_v2.txt 2>&1
cat gpurun_out/r06_timing_v2.txt

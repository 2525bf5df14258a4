set -x
python bench.py > gpurun_out/r06_bench_full_1.json 2> gpurun_out/r06_bench_full_1.log
tail -5 gpurun_out/r06_bench_full_1.log
python tools/ab.py --workloads c5,c6,c3,c2 --repeat 2 --steps 10 --tag sdwa tree > gpurun_out/r06_ab_sdwa.log 2>&1
cat gpurun_out/ab_sdwa.txt

#!/bin/bash
O=gpurun_out/r02k; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?" >> $O/bench_default.err
tail -30 $O/pytest_gpu.txt; tail -12 $O/bench_default.err

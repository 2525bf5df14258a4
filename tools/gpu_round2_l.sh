#!/bin/bash
O=gpurun_out/r02l; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
timeout 2400 tools/measure_round.sh r02 > $O/measure.txt 2>&1
tail -5 $O/pytest_gpu.txt; cat $O/measure.txt

#!/bin/bash
# GPU suite only (no measurement round)
O=gpurun_out/r02t; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -22 $O/pytest_gpu.txt

#!/bin/bash
# GPU suite only (no measurement round)
O=gpurun_out/r02t; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt

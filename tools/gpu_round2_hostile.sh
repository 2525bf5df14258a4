#!/bin/bash
# throughput of the hostile pattern sets (printed by the tests themselves) + the whole GPU suite
O=gpurun_out/r02hostile; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q -s -k "snort_length or every_position" > $O/out.txt 2>&1
grep -A1 "input GB/s" $O/out.txt; tail -2 $O/out.txt
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt

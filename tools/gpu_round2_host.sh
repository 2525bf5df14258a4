#!/bin/bash
O=gpurun_out/r02host; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q -x -k "match_from_host" > $O/pytest_host.txt 2>&1; tail -4 $O/pytest_host.txt

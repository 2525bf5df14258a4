#!/bin/bash
O=gpurun_out/r02small; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/small_input_latency.py > $O/small.txt 2>&1; tail -25 $O/small.txt

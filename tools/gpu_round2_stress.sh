#!/bin/bash
O=gpurun_out/r02stress; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python tools/stress_fuzz.py 1000 120 > $O/stress.txt 2>&1; echo "rc $?" >> $O/stress.txt
tail -8 $O/stress.txt
